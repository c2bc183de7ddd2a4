// PPM input/output of the drop-in surface (Code/IO/Image.cpp): textures are read from P3 or P6
// files, frames are written as ASCII P3 with five pixels to a line.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "mirt_scene.h"

Image::Image(int h, int w, int maxv, std::string magic)
    : height(h), width(w), max_value(maxv), magic_number(std::move(magic)) {
  pixels.assign(h, std::vector<Pixel>(w));
}

namespace {
// Next header integer; a `#` comment may follow a value or stand on a line of its own
// (Image.cpp:89-115).
int header_int(std::ifstream &in) {
  std::string line;
  int v = 0;
  for (;;) {
    in >> v;
    if (in.good()) {
      in >> std::ws;
      if (in.peek() == '#') std::getline(in, line);
      return v;
    }
    in.clear();
    if (!std::getline(in, line)) return v;
    if (line.empty() || line[0] != '#') {
      std::istringstream s(line);
      s >> v;
      return v;
    }
  }
}
}  // namespace

Image::Image(const std::string &filepath) {
  std::ifstream in(filepath, std::ios::binary);
  if (!in) {
    std::cerr << "[ERROR] Error: Could not open file path=" << filepath << "\n";
    return;
  }
  in >> magic_number;
  if (magic_number != "P3" && magic_number != "P6") {
    std::cerr << "[ERROR] Error: Invalid PPM format format=" << magic_number << "\n";
    return;
  }
  width = header_int(in);
  height = header_int(in);
  max_value = header_int(in);
  if (width <= 0 || height <= 0) {
    width = height = 0;
    return;
  }
  pixels.assign(height, std::vector<Pixel>(width));
  if (magic_number == "P3") {
    for (auto &row : pixels)
      for (auto &p : row) {
        int r = 0, g = 0, b = 0;
        in >> r >> g >> b;
        p = Pixel(static_cast<double>(r) / max_value, static_cast<double>(g) / max_value,
                  static_cast<double>(b) / max_value);
      }
  } else {
    // the header reader has already skipped white space after max_value; the reference skips
    // once more here (Image.cpp:146), so leading raster bytes that are ASCII blanks are lost
    in >> std::ws;
    std::vector<unsigned char> row(static_cast<std::size_t>(width) * 3);
    for (auto &line : pixels) {
      std::fill(row.begin(), row.end(), 0);
      in.read(reinterpret_cast<char *>(row.data()), static_cast<std::streamsize>(row.size()));
      for (int x = 0; x < width; ++x)
        line[x] = Pixel(static_cast<double>(row[3 * x]) / max_value, static_cast<double>(row[3 * x + 1]) / max_value,
                        static_cast<double>(row[3 * x + 2]) / max_value);
    }
  }
}

void Image::write(const std::string &filepath) const {
  const bool binary = magic_number == "P6";
  std::FILE *f = std::fopen(filepath.c_str(), binary ? "wb" : "w");
  if (!f) {
    std::cerr << "[ERROR] Error: Could not open file for writing path=" << filepath << "\n";
    return;
  }
  auto quant = [this](double v) {  // clamp to [0,1], round half away from zero (Image.cpp:123-128)
    v = std::clamp(v, 0.0, 1.0);
    return static_cast<unsigned char>(std::lround(v * max_value));
  };
  std::fprintf(f, "%s\n%d %d\n%d\n", magic_number.c_str(), width, height, max_value);
  std::string out;
  out.reserve(static_cast<std::size_t>(width) * 13 + 16);
  for (int y = 0; y < height; ++y) {
    out.clear();
    for (int x = 0; x < width; ++x) {
      const Pixel &p = pixels[y][x];
      if (binary) {
        out.push_back(static_cast<char>(quant(p.x)));
        out.push_back(static_cast<char>(quant(p.y)));
        out.push_back(static_cast<char>(quant(p.z)));
      } else {
        char buf[16];
        out.append(buf, std::snprintf(buf, sizeof buf, "%d %d %d ", quant(p.x), quant(p.y), quant(p.z)));
        if ((x + 1) % 5 == 0) out.push_back('\n');
      }
    }
    if (!binary) out.push_back('\n');
    std::fwrite(out.data(), 1, out.size(), f);
  }
  std::fclose(f);
}
