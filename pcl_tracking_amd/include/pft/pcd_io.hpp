// pcd_io.hpp -- minimal reader for the on-disk format the reference's model-creation nodes write
// (pcl::PCDWriter::write<PointXYZRGBA>(path, cloud, false), /root/reference/src/create_model.cpp:219-222) and
// that auto_tracking.cpp once loaded directly (pcl::io::loadPCDFile, :741): PCD v0.7, DATA ascii, binary or binary_compressed,
// fields x y z and rgba (TYPE U) or rgb (TYPE F, packed bits).  Host-side I/O only; other
// fields are skipped.  DATA binary_compressed (LZF; what pcl::PCDWriter::writeBinaryCompressed and pcl_convert_pcd_ascii_binary
// produce) is read too.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "pft/particle_filter_tracker.hpp"

namespace pft {
namespace io {

// LZF (Marc Lehmann's liblzf, the codec of PCD "binary_compressed"): a control byte below 32 starts a run of ctrl + 1
// literal bytes; otherwise its top three bits are a match length (7 = one more length byte follows), its low five bits
// and the next byte the distance - 1 of the match, which copies length + 2 bytes (the regions may overlap).  Returns
// the number of bytes produced, 0 on malformed input or if they do not fit.
inline size_t lzfDecompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    const unsigned ctrl = in[ip++];
    if (ctrl < 32u) {
      const size_t run = ctrl + 1u;
      if (ip + run > in_len || op + run > out_len) return 0;
      std::memcpy(out + op, in + ip, run);
      ip += run;
      op += run;
    } else {
      size_t len = ctrl >> 5;
      if (len == 7u) {
        if (ip >= in_len) return 0;
        len += in[ip++];
      }
      if (ip >= in_len) return 0;
      const size_t dist = (((size_t)ctrl & 0x1fu) << 8) + in[ip++] + 1u;
      len += 2u;
      if (dist > op || op + len > out_len) return 0;
      for (size_t k = 0; k < len; k++, op++) out[op] = out[op - dist];  // byte by byte: the match may overlap its own output
    }
  }
  return op;
}

// returns 0 on success, -1 on failure (as pcl::io::loadPCDFile does)
inline int loadPCDFile(const std::string& path, PointCloud<PointXYZRGBA>& cloud) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return -1;
  std::vector<std::string> fields;
  std::vector<int> sizes, counts;
  std::vector<char> types;
  size_t points = 0, width = 0, height = 1;
  std::string data_mode;
  char line[4096];
  bool have_points = false;
  while (std::fgets(line, sizeof(line), f)) {
    if (line[0] == '#') continue;
    std::istringstream is(line);
    std::string key;
    is >> key;
    if (key == "FIELDS" || key == "COLUMNS") {
      std::string s;
      while (is >> s) fields.push_back(s);
    } else if (key == "SIZE") {
      int v;
      while (is >> v) sizes.push_back(v);
    } else if (key == "TYPE") {
      char c;
      while (is >> c) types.push_back(c);
    } else if (key == "COUNT") {
      int v;
      while (is >> v) counts.push_back(v);
    } else if (key == "WIDTH") {
      is >> width;
    } else if (key == "HEIGHT") {
      is >> height;
    } else if (key == "POINTS") {
      is >> points;
      have_points = true;
    } else if (key == "DATA") {
      is >> data_mode;
      break;
    }
  }
  const size_t nf = fields.size();
  if (!nf || sizes.size() != nf || types.size() != nf || data_mode.empty()) {
    std::fclose(f);
    return -1;
  }
  if (counts.empty()) counts.assign(nf, 1);
  if (!have_points) points = width * height;
  int ix = -1, iy = -1, iz = -1, ic = -1;
  std::vector<size_t> offs(nf, 0);
  size_t stride = 0;
  for (size_t k = 0; k < nf; k++) {
    offs[k] = stride;
    stride += (size_t)sizes[k] * (size_t)counts[k];
    if (fields[k] == "x") ix = (int)k;
    if (fields[k] == "y") iy = (int)k;
    if (fields[k] == "z") iz = (int)k;
    if (fields[k] == "rgba" || fields[k] == "rgb") ic = (int)k;
  }
  if (ix < 0 || iy < 0 || iz < 0 || sizes[ix] != 4 || sizes[iy] != 4 || sizes[iz] != 4 || (ic >= 0 && sizes[ic] != 4)) {
    std::fclose(f);
    return -1;
  }
  cloud.points.assign(points, PointXYZRGBA());
  bool dense = true;
  if (data_mode == "ascii") {
    for (size_t i = 0; i < points; i++) {
      if (!std::fgets(line, sizeof(line), f)) {
        std::fclose(f);
        return -1;
      }
      std::vector<const char*> tok;
      for (char* t = std::strtok(line, " \t\r\n"); t; t = std::strtok(nullptr, " \t\r\n")) tok.push_back(t);
      size_t ti = 0;
      PointXYZRGBA& q = cloud.points[i];
      for (size_t k = 0; k < nf; k++) {
        for (int c = 0; c < counts[k]; c++) {
          if (ti >= tok.size()) {
            std::fclose(f);
            return -1;
          }
          const char* s = tok[ti++];
          if ((int)k == ic && types[k] != 'F') {
            q.rgba = (uint32_t)std::strtoul(s, nullptr, 10);
          } else {
            const float v = std::strtof(s, nullptr);  // accepts "nan"
            if ((int)k == ix) q.x = v;
            else if ((int)k == iy) q.y = v;
            else if ((int)k == iz) q.z = v;
            else if ((int)k == ic) std::memcpy(&q.rgba, &v, 4);  // rgb as a float carrying the packed bits
          }
        }
      }
      if (q.x != q.x || q.y != q.y || q.z != q.z) dense = false;
    }
  } else if (data_mode == "binary") {
    std::vector<unsigned char> row(stride);
    for (size_t i = 0; i < points; i++) {
      if (std::fread(row.data(), 1, stride, f) != stride) {
        std::fclose(f);
        return -1;
      }
      PointXYZRGBA& q = cloud.points[i];
      std::memcpy(&q.x, row.data() + offs[ix], 4);
      std::memcpy(&q.y, row.data() + offs[iy], 4);
      std::memcpy(&q.z, row.data() + offs[iz], 4);
      if (ic >= 0) std::memcpy(&q.rgba, row.data() + offs[ic], 4);
      if (q.x != q.x || q.y != q.y || q.z != q.z) dense = false;
    }
  } else if (data_mode == "binary_compressed") {
    // pcl::PCDWriter::writeBinaryCompressed: u32 compressed size, u32 uncompressed size, LZF stream; the uncompressed
    // buffer holds the fields one after the other (all x, then all y, ...), not the points
    uint32_t csize = 0, usize = 0;
    if (std::fread(&csize, 4, 1, f) != 1 || std::fread(&usize, 4, 1, f) != 1 || (size_t)usize != stride * points) {
      std::fclose(f);
      return -1;
    }
    std::vector<unsigned char> comp(csize), buf(usize);
    if (csize && std::fread(comp.data(), 1, csize, f) != csize) {
      std::fclose(f);
      return -1;
    }
    if (lzfDecompress(comp.data(), csize, buf.data(), usize) != usize) {
      std::fclose(f);
      return -1;
    }
    std::vector<size_t> block(nf, 0);  // start of every field's block
    size_t acc = 0;
    for (size_t k = 0; k < nf; k++) {
      block[k] = acc;
      acc += (size_t)sizes[k] * (size_t)counts[k] * points;
    }
    for (size_t i = 0; i < points; i++) {
      PointXYZRGBA& q = cloud.points[i];
      std::memcpy(&q.x, buf.data() + block[ix] + 4 * i * (size_t)counts[ix], 4);
      std::memcpy(&q.y, buf.data() + block[iy] + 4 * i * (size_t)counts[iy], 4);
      std::memcpy(&q.z, buf.data() + block[iz] + 4 * i * (size_t)counts[iz], 4);
      if (ic >= 0) std::memcpy(&q.rgba, buf.data() + block[ic] + 4 * i * (size_t)counts[ic], 4);
      if (q.x != q.x || q.y != q.y || q.z != q.z) dense = false;
    }
  } else {
    std::fclose(f);
    return -1;
  }
  std::fclose(f);
  cloud.width = (uint32_t)(width ? width : points);
  cloud.height = (uint32_t)height;
  cloud.is_dense = dense;
  return 0;
}

}  // namespace io
}  // namespace pft
