// rusty_sr -- host CLI with the argv surface of millardjn/rusty_sr v1
// (reference src/main.rs:33-178), driving the MI355X engine through libsrhip's C ABI.
//
//   rusty_sr <INPUT_FILE> <OUTPUT_FILE> [-p imagenet|imagenetlinear|anime|bilinear] [-c FILE] [-d]
//
// Differences from the reference, all outside the hot path: own codecs (PNG over zlib, baseline + progressive JPEG, GIF,
// TIFF, TGA, ICO, PPM/PGM/PBM, BMP in; PNG, JPEG, BMP, PPM out by extension -- of what the reference's `image` crate reads
// only WebP is missing), the `train` sub-command
// is not part of this build, and three extra options that cannot collide with the
// reference's (-p -c -d): --device N, --precision f32|split_f16, --timing.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cctype>
#include <cstring>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "../../include/srhip.h"
#include "png.hpp"

// The three parameter sets the reference embeds with include_bytes! (main.rs:26-28)
#define EMBED(sym, path)                                                                    \
    asm(".section .rodata\n.global " #sym "_begin\n" #sym "_begin:\n.incbin \"" path "\"\n" \
        ".global " #sym "_end\n" #sym "_end:\n.byte 0\n.previous\n");                      \
    extern "C" const unsigned char sym##_begin[], sym##_end[];
EMBED(imagenet_rsr, SR_RES_DIR "/imagenet.rsr")
EMBED(imagenetlinear_rsr, SR_RES_DIR "/imagenetlinear.rsr")
EMBED(anime_rsr, SR_RES_DIR "/anime.rsr")

namespace {

const char* kUsage =
    "Rusty SR v0.1.1 (MI355X engine)\n"
    "A convolutional neural network trained to upscale images\n\n"
    "USAGE:\n    rusty_sr [FLAGS] [OPTIONS] <INPUT_FILE> <OUTPUT_FILE>\n    rusty_sr train ...   (not part of this build)\n\n"
    "FLAGS:\n    -d, --downsample    Perform downscaling rather than upscaling\n    -h, --help          Prints help information\n"
    "    -V, --version       Prints version information\n        --timing        Print device / transfer times on stderr\n\n"
    "OPTIONS:\n    -c, --custom <PARAMETER_FILE>    Sets a custom parameter file (.rsr) to use with the neural net\n"
    "    -p, --parameters <PARAMETERS>    Sets which built-in parameters to use with the neural net [values: imagenet,\n"
    "                                     imagenetlinear, anime, bilinear]\n"
    "        --device <N>                 HIP device index [default: 0]\n"
    "        --devices <N,N,...>          spread one image over several GPUs (row shares, halo rows from the image)\n"
    "        --precision <MODE>           f32 (exact) or split_f16 (2x faster, same 1e-4 parity bar) [default: f32]\n\n"
    "ARGS:\n    <INPUT_FILE>     Sets the input image to upscale\n    <OUTPUT_FILE>    Sets the output file to write/overwrite (.png recommended)\n";

[[noreturn]] void die(const std::string& msg, int code = 1) {
    fprintf(stderr, "error: %s\n", msg.c_str());
    exit(code);
}
[[noreturn]] void usage_error(const std::string& msg) {
    fprintf(stderr, "error: %s\n\nUSAGE:\n    rusty_sr [FLAGS] [OPTIONS] <INPUT_FILE> <OUTPUT_FILE>\n\nFor more information try --help\n", msg.c_str());
    exit(2);
}

std::vector<float> decode_rsr(const unsigned char* blob, size_t len) {
    size_t n = 0;
    if (sr_rsr_decode(blob, len, nullptr, 0, &n) != SR_OK) die("ByteVec conversion failed");  // main.rs:138
    std::vector<float> p(n);
    if (sr_rsr_decode(blob, len, p.data(), n, &n) != SR_OK) die("ByteVec conversion failed");
    return p;
}

}  // namespace

int main(int argc, char** argv) {
    std::vector<std::string> pos;
    std::string parameters, custom, precision = "f32";
    bool has_p = false, has_c = false, downsample = false, timing = false;
    int device = 0;
    std::vector<int> devices;
    if (argc >= 2 && !strcmp(argv[1], "train"))  // main.rs:119-121
        die("the `train` sub-command is not part of this build (the MI355X engine covers the upscale path only)", 2);
    for (int k = 1; k < argc; ++k) {
        const std::string a = argv[k];
        auto value = [&](const char* name) -> std::string {
            if (k + 1 >= argc) usage_error(std::string("The argument '") + name + "' requires a value but none was supplied");
            return argv[++k];
        };
        if (a == "-h" || a == "--help") { fputs(kUsage, stdout); return 0; }
        else if (a == "-V" || a == "--version") { puts("Rusty SR v0.1.1"); return 0; }
        else if (a == "-d" || a == "--downsample") downsample = true;
        else if (a == "--timing") timing = true;
        else if (a == "-p" || a == "--parameters") { parameters = value("--parameters <PARAMETERS>"); has_p = true; }
        else if (a.rfind("--parameters=", 0) == 0) { parameters = a.substr(13); has_p = true; }
        else if (a == "-c" || a == "--custom") { custom = value("--custom <PARAMETER_FILE>"); has_c = true; }
        else if (a.rfind("--custom=", 0) == 0) { custom = a.substr(9); has_c = true; }
        else if (a == "--device") device = atoi(value("--device <N>").c_str());
        else if (a == "--devices") {
            const std::string list = value("--devices <N,N,...>");
            for (size_t pos0 = 0; pos0 <= list.size();) {
                const size_t comma = std::min(list.find(',', pos0), list.size());
                if (comma == pos0 || !isdigit((unsigned char)list[pos0])) usage_error("'" + list + "' isn't a valid value for '--devices <N,N,...>'");
                devices.push_back(atoi(list.substr(pos0, comma - pos0).c_str()));
                pos0 = comma + 1;
            }
        }
        else if (a == "--precision") precision = value("--precision <MODE>");
        else if (a.size() > 1 && a[0] == '-') usage_error("Found argument '" + a + "' which wasn't expected, or isn't valid in this context");
        else pos.push_back(a);
    }
    // clap rules of the reference: possible_values (main.rs:54), conflicts (main.rs:58,66), required positionals
    if (has_p && parameters != "imagenet" && parameters != "imagenetlinear" && parameters != "anime" && parameters != "bilinear")
        usage_error("'" + parameters + "' isn't a valid value for '--parameters <PARAMETERS>'\n\t[values: anime, bilinear, imagenet, imagenetlinear]");
    if (has_c && has_p) usage_error("The argument '--custom <PARAMETER_FILE>' cannot be used with '--parameters <PARAMETERS>'");
    if (downsample && (has_p || has_c)) usage_error("The argument '--downsample' cannot be used with '--parameters <PARAMETERS>' or '--custom <PARAMETER_FILE>'");
    if (pos.size() < 2) usage_error("The following required arguments were not provided:\n    <INPUT_FILE>\n    <OUTPUT_FILE>");
    if (pos.size() > 2) usage_error("Found argument '" + pos[2] + "' which wasn't expected, or isn't valid in this context");
    if (precision != "f32" && precision != "split_f16") usage_error("'" + precision + "' isn't a valid value for '--precision <MODE>'");

    // ---- parameters + graph (main.rs:133-158), same progress text
    std::vector<float> params;
    int graph = SR_GRAPH_SR_NET;
    if (has_c) {
        FILE* f = fopen(custom.c_str(), "rb");
        if (!f) die("Error opening parameter file");  // main.rs:134
        std::vector<unsigned char> data;
        unsigned char tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) data.insert(data.end(), tmp, tmp + n);
        fclose(f);
        printf("Upscaling using custom neural net parameters...");
        params = decode_rsr(data.data(), data.size());
    } else if (downsample) {
        printf("Downsampling using average pooling of linear RGB values...");
        graph = SR_GRAPH_DOWNSAMPLE;
    } else if (!has_p || parameters == "imagenet") {
        printf("Upscaling using imagenet neural net parameters...");
        params = decode_rsr(imagenet_rsr_begin, imagenet_rsr_end - imagenet_rsr_begin);
    } else if (parameters == "imagenetlinear") {
        printf("Upscaling using linear loss imagenet neural net parameters...");
        params = decode_rsr(imagenetlinear_rsr_begin, imagenetlinear_rsr_end - imagenetlinear_rsr_begin);
    } else if (parameters == "anime") {
        printf("Upscaling using anime neural net parameters...");
        params = decode_rsr(anime_rsr_begin, anime_rsr_end - anime_rsr_begin);
    } else {
        printf("Upscaling using bilinear interpolation...");
        graph = SR_GRAPH_BILINEAR;
    }
    fflush(stdout);

    if (devices.empty() || graph != SR_GRAPH_SR_NET) devices.assign(1, devices.empty() ? device : devices[0]);
    {   // `.save()` picks the container from the extension (main.rs:175): refuse an unknown one before any work is spent on the picture
        const size_t dot = pos[1].find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : pos[1].substr(dot + 1);
        for (auto& ch : ext) ch = (char)tolower((unsigned char)ch);
        if (ext != "png" && ext != "jpg" && ext != "jpeg" && ext != "bmp" && ext != "ppm")
            die("Could not write output file (this build writes .png, .jpg, .bmp and .ppm)");
    }
    // The input file decodes on a second thread while this one brings up the device and the contexts (HIP start-up
    // and the weight upload take longer than a 1080p PNG); failures are then reported in the reference's order --
    // the graph first (main.rs:160-162), the image after it (main.rs:164).
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
    const clk::time_point t_start = clk::now();
    srpng::Image in;
    std::string err;
    bool decoded = false;
    double t_decode = 0;
    std::thread decoder([&] {
        const clk::time_point t = clk::now();
        decoded = srpng::decode_image_file(pos[0], in, err);
        t_decode = ms_since(t);
    });
    // The file's header already says how large the output is (probe_image_size): a third thread page-locks it -- 15 ms at
    // 1080p, 60 ms at 4K, and independent of the context -- while this one creates the context, lets the library allocate and
    // warm what the call will need (sr_reserve_rgba8), and the decoder is still busy.
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    int pw = 0, ph = 0;
    const bool sized = srpng::probe_image_size(pos[0], pw, ph) && !(graph == SR_GRAPH_DOWNSAMPLE && (pw < 3 || ph < 3));
    std::thread pinner([&] {
        if (!sized) return;
        const size_t bytes = graph == SR_GRAPH_DOWNSAMPLE ? (size_t)(pw / 3) * (ph / 3) * 4 : (size_t)pw * 3 * ph * 3 * 4;
        if (sr_host_alloc(&pinned, bytes) == SR_OK) pinned_bytes = bytes; else pinned = nullptr;
    });
    std::vector<sr_ctx*> ctxs(devices.size(), nullptr);
    int rc = SR_OK;
    for (size_t k = 0; k < devices.size() && rc == SR_OK; ++k) {
        rc = sr_create_graph(&ctxs[k], graph, params.empty() ? nullptr : params.data(), params.size(), SR_FACTOR, devices[k]);
        if (rc == SR_OK && graph == SR_GRAPH_SR_NET) rc = sr_set_precision(ctxs[k], precision == "f32" ? SR_PRECISION_F32 : SR_PRECISION_SPLIT_F16);
    }
    const double t_create = ms_since(t_start);
    double t_prep = 0;
    {
        const clk::time_point t = clk::now();
        if (rc == SR_OK && sized && ctxs.size() == 1) (void)sr_reserve_rgba8(ctxs[0], 4, 1, ph, pw);  // best effort: the real call reports errors
        pinner.join();
        t_prep = ms_since(t);
    }
    decoder.join();
    if (rc != SR_OK) die(sr_strerror(rc));  // SR_E_PARAM_COUNT carries the text of main.rs:162
    sr_ctx* ctx = ctxs[0];
    if (!decoded) die("Error opening input image file. (" + err + ")");  // main.rs:164
    const double t_ready = ms_since(t_start);
    if (graph == SR_GRAPH_DOWNSAMPLE && (in.w < 3 || in.h < 3)) die("input image is smaller than one 3x3 pooling block");

    const int ow = graph == SR_GRAPH_DOWNSAMPLE ? in.w / 3 : in.w * 3, oh = graph == SR_GRAPH_DOWNSAMPLE ? in.h / 3 : in.h * 3;
    // page-locked output pixels: the download then runs at PCIe rate under the kernels of the next band
    const size_t out_bytes = (size_t)ow * oh * 4;
    const clk::time_point t_al = clk::now();
    std::vector<uint8_t> pageable;
    if (pinned && pinned_bytes != out_bytes) { sr_host_free(pinned); pinned = nullptr; }  // the header lied
    if (!pinned && sr_host_alloc(&pinned, out_bytes) != SR_OK) { pinned = nullptr; pageable.resize(out_bytes); }
    uint8_t* out = pinned ? (uint8_t*)pinned : pageable.data();
    // img_to_data + graph.forward + data_to_img(..).to_rgba(), fused on the device (main.rs:168-175)
    const double t_alloc = ms_since(t_al);
    const clk::time_point t_up = clk::now();
    rc = ctxs.size() > 1 ? sr_upscale_rgba8_multi(ctxs.data(), (int)ctxs.size(), in.rgba.data(), 4, in.h, in.w, out)
                         : sr_upscale_rgba8(ctx, in.rgba.data(), 4, 1, in.h, in.w, out);
    if (rc != SR_OK) die(std::string(sr_strerror(rc)) + (rc == SR_E_HIP ? " (hipError " + std::to_string(sr_last_hip_error(ctx)) + ")" : ""));
    const double t_upscale = ms_since(t_up);
    if (timing) {
        double tot = 0, h2d = 0, d2h = 0;
        sr_last_timing(ctx, &tot, nullptr, &h2d, &d2h);
        fprintf(stderr, "\n[timing] %dx%d -> %dx%d: kernels %.3f ms, h2d %.3f ms, d2h %.3f ms\n", in.w, in.h, ow, oh, tot, h2d, d2h);
    }
    printf(" Writing file...");
    fflush(stdout);
    const clk::time_point t_enc = clk::now();
    if (!srpng::encode_image_file(pos[1], out, ow, oh, err)) die("Could not write output file (" + err + ")");  // main.rs:175
    const double t_encode = ms_since(t_enc);
    puts(" Done");
    if (timing)  // t_cli of SURVEY.md 8(d): everything this process did, by phase (decode and device start-up overlap)
        fprintf(stderr, "[timing] wall: decode %.1f ms || device + contexts %.1f ms, reserve (|| page-locking the output) %.1f ms -> ready at %.1f ms; "
                        "late allocations %.1f ms; upscale call %.1f ms; encode + write %.1f ms; total %.1f ms\n", t_decode, t_create, t_prep, t_ready,
                t_alloc, t_upscale, t_encode, ms_since(t_start));
    sr_host_free(pinned);
    for (sr_ctx* c : ctxs) sr_destroy(c);
    return 0;
}
