#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Writes fixtures/ASCII/Test4_ties.txt: the bundled Test4 (a glass cylinder standing
on the ground plane and other coincident surfaces, where two hits lie at exactly the same distance) with
600 small opaque spheres strewn around it, which takes the scene past the 512 primitives at which the
engine picks its large-scene walk by itself.  Fixed seed: the same file every time."""
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fixtures", "ASCII", "Test4.txt")
DST = os.path.join(ROOT, "fixtures", "ASCII", "Test4_ties.txt")
EXTRA = 600


def main():
    lines = open(SRC).read().split("\n")
    at = next(i for i, l in enumerate(lines) if l.startswith("SPHERES "))
    n = int(lines[at].split()[1])
    end = next(i for i in range(at + 1, len(lines)) if lines[i].startswith("CUBES "))
    rnd = random.Random(4)
    extra = []
    for k in range(EXTRA):
        r = rnd.uniform(0.03, 0.09)
        x, y = rnd.uniform(-7.0, 7.0), rnd.uniform(-7.0, 7.0)
        z = r if k % 3 else rnd.uniform(0.2, 4.0)  # two in three rest ON the ground plane: one more tie each
        c = [rnd.uniform(0.2, 1.0) for _ in range(3)]
        refl = 0.6 if k % 7 == 0 else 0.0
        extra += [f"name Filler_{k}", f"location {x:.6f} {y:.6f} {z:.6f}", "rotation 0.0 0.0 0.0", f"scale {r:.6f} {r:.6f} {r:.6f}",
                  "visible 1", f"material_diffuse {c[0]:.4f} {c[1]:.4f} {c[2]:.4f}", "material_specular 0.5 0.5 0.5",
                  f"material_ambient {c[0] / 10:.4f} {c[1] / 10:.4f} {c[2] / 10:.4f}", "material_shininess 48.0",
                  "material_glossiness 1.0", f"material_reflectivity {refl}", "material_transparency 0.0",
                  "material_refractive_index 1.5"]
    out = lines[:at] + [f"SPHERES {n + EXTRA}"] + lines[at + 1:end] + extra + lines[end:]
    with open(DST, "w") as f:
        f.write("\n".join(out))
    print("wrote", DST, n + EXTRA, "spheres")


if __name__ == "__main__":
    main()
