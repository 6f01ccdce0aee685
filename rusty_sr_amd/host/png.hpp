// png.hpp -- minimal PNG codec over zlib for the rusty_sr host CLI.
// Stands in for the `image` crate calls of the reference (image::open main.rs:164,
// DynamicImage::to_rgba().save main.rs:175).  Decodes non-interlaced and Adam7
// PNGs of colour types 0/2/3/4/6 at 1..16 bits to RGBA8 (16-bit samples keep their
// high byte; no gamma / colour management, like the reference); encodes RGBA8.  Plus the other containers the CLI
// reads (JPEG baseline + progressive, GIF, TIFF, TGA, ICO, PNM, BMP) and writes (PNG, JPEG, BMP, PPM).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace srpng {
struct Image { int w = 0, h = 0; std::vector<uint8_t> rgba; };
bool decode_file(const std::string& path, Image& out, std::string& err);
bool decode_memory(const uint8_t* data, size_t len, Image& out, std::string& err);
// zlevel > 0 (default): run-length matches + dynamic Huffman, the fast and -- on filtered continuous-tone pixels -- also
// the smaller choice; zlevel == 0: zlib's general matcher at level 3
bool encode_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err, int zlevel = -1);
unsigned usable_cpus();  // affinity mask capped by the cgroup CPU quota
// rle_deflate.cpp: raw deflate with run-length matches + dynamic Huffman only (what the PNG encoder uses); 0 = cap too small
size_t rle_deflate_bound(size_t n);
size_t rle_deflate(const uint8_t* src, size_t n, bool last, uint8_t* dst, size_t cap);
// baseline JPEG (jpeg.cpp); binary PPM / PGM and uncompressed 24 / 32-bit BMP (png.cpp)
bool decode_jpeg_memory(const uint8_t* data, size_t len, Image& out, std::string& err);
// image::open stand-in: picks the decoder from the file's magic bytes (PNG, JPEG, GIF, TIFF, BMP, ICO, PPM/PGM/PBM;
// formats.cpp has GIF / TIFF / ICO / TGA) -- TGA, which has none, from the .tga extension
bool decode_image_file(const std::string& path, Image& out, std::string& err);
// width / height from the first bytes of the file (PNG, BMP, GIF, PNM, JPEG with its frame header in the first 64 KB) without
// decoding it: lets the CLI size its buffers while the decoder still runs.  false: unknown (decode will tell).
bool probe_image_size(const std::string& path, int& w, int& h);
// `.save(path)` stand-in (reference main.rs:175: the image crate picks the container from the extension):
// .png (RGBA8), .jpg / .jpeg (baseline, quality 75, alpha dropped), .bmp (32-bit), .ppm (binary P6, alpha dropped)
bool encode_image_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err);
bool encode_jpeg_file(const std::string& path, const uint8_t* rgba, int w, int h, std::string& err, int quality = 75);
}  // namespace srpng

extern "C" {
// C surface used by tests/test_host_png.py through ctypes
int srpng_decode_rgba8(const char* path, int* w, int* h, uint8_t** rgba);  // caller frees with srpng_free
int srpng_decode_any_rgba8(const char* path, int* w, int* h, uint8_t** rgba);  // PNG / JPEG / PPM / BMP by magic
int srpng_encode_rgba8(const char* path, const uint8_t* rgba, int w, int h);
int srpng_encode_any_rgba8(const char* path, const uint8_t* rgba, int w, int h);  // container by extension: png / jpg / bmp / ppm
int srpng_probe_size(const char* path, int* w, int* h);  // header only; -1 if it cannot tell
void srpng_free(uint8_t* p);
}
