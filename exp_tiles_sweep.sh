for g in 256 512 768 1024 1536 2048 4096; do echo "consumer blocks $g"; MIRT_CONSUMER_BLOCKS=$g timeout -k 10 200 python3 exp_tiles.py | grep -E "world (1|8) rank 0" || exit 1; done
