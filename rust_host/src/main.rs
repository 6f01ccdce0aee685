//! rusty_sr -- Rust host over libsrhip (the MI355X engine).
//!
//!     rusty_sr <INPUT_FILE> <OUTPUT_FILE> [-p imagenet|imagenetlinear|anime|bilinear] [-c FILE] [-d]
//!
//! Same arguments, progress text and failure text as millardjn/rusty_sr v1; decoding and encoding
//! of image files stay with the `image` crate as in the reference, everything between the decoded
//! pixels and the pixels to encode runs on the GPU.  `train` is not part of this host.
//! This file is not compiled in the repository's image (no Rust toolchain); the C++ twin
//! `rusty_sr_amd/host/main.cpp` is what the tests drive, and `tests/test_rust_host.py` keeps the
//! two in step (same option names, same strings, every FFI symbol exported).
extern crate image;

mod srhip;

use std::env;
use std::fs::File;
use std::io::{stdout, Read, Write};
use std::path::Path;
use std::process::exit;

use srhip::Engine;

static IMAGENET: &'static [u8] = include_bytes!("../../rusty_sr_amd/res/imagenet.rsr");
static IMAGENETLINEAR: &'static [u8] = include_bytes!("../../rusty_sr_amd/res/imagenetlinear.rsr");
static ANIME: &'static [u8] = include_bytes!("../../rusty_sr_amd/res/anime.rsr");

const BUILTIN: [&'static str; 4] = ["imagenet", "imagenetlinear", "anime", "bilinear"];

struct Options {
    input: String,
    output: String,
    parameters: Option<String>,
    custom: Option<String>,
    downsample: bool,
    device: i32,
    split_f16: bool,
    timing: bool,
}

fn usage_error(msg: &str) -> ! {
    let _ = writeln!(std::io::stderr(),
        "error: {}\n\nUSAGE:\n    rusty_sr [FLAGS] [OPTIONS] <INPUT_FILE> <OUTPUT_FILE>\n\nFor more information try --help", msg);
    exit(2)
}

fn die(msg: &str) -> ! {
    let _ = writeln!(std::io::stderr(), "error: {}", msg);
    exit(1)
}

fn parse_args() -> Options {
    let mut o = Options { input: String::new(), output: String::new(), parameters: None, custom: None,
                          downsample: false, device: 0, split_f16: false, timing: false };
    let mut positional: Vec<String> = Vec::new();
    let mut args = env::args().skip(1);
    while let Some(a) = args.next() {
        let mut value = |name: &str| -> String {
            match args.next() {
                Some(v) => v,
                None => usage_error(&format!("The argument '{}' requires a value but none was supplied", name)),
            }
        };
        match a.as_str() {
            "train" if positional.is_empty() => {
                let _ = writeln!(std::io::stderr(), "error: the `train` sub-command is not part of this build");
                exit(2)
            }
            "-h" | "--help" => {
                println!("Rusty SR v0.1.1 (MI355X engine)\nUSAGE:\n    rusty_sr [-d] [-p PARAMETERS | -c PARAMETER_FILE] \
                          [--device N] [--precision f32|split_f16] [--timing] <INPUT_FILE> <OUTPUT_FILE>");
                exit(0)
            }
            "-V" | "--version" => {
                println!("Rusty SR v0.1.1");
                exit(0)
            }
            "-d" | "--downsample" => o.downsample = true,
            "--timing" => o.timing = true,
            "-p" | "--parameters" => o.parameters = Some(value("--parameters <PARAMETERS>")),
            "-c" | "--custom" => o.custom = Some(value("--custom <PARAMETER_FILE>")),
            "--device" => o.device = value("--device <N>").parse().unwrap_or_else(|_| usage_error("--device takes an integer")),
            "--precision" => {
                let v = value("--precision <MODE>");
                match v.as_str() {
                    "f32" => o.split_f16 = false,
                    "split_f16" => o.split_f16 = true,
                    _ => usage_error(&format!("'{}' isn't a valid value for '--precision <MODE>'", v)),
                }
            }
            s if s.len() > 1 && s.starts_with('-') => {
                usage_error(&format!("Found argument '{}' which wasn't expected, or isn't valid in this context", s))
            }
            _ => positional.push(a.clone()),
        }
    }
    if let Some(ref p) = o.parameters {
        if !BUILTIN.contains(&p.as_str()) {
            usage_error(&format!("'{}' isn't a valid value for '--parameters <PARAMETERS>'\n\t[values: anime, bilinear, imagenet, imagenetlinear]", p));
        }
    }
    if o.custom.is_some() && o.parameters.is_some() {
        usage_error("The argument '--custom <PARAMETER_FILE>' cannot be used with '--parameters <PARAMETERS>'");
    }
    if o.downsample && (o.custom.is_some() || o.parameters.is_some()) {
        usage_error("The argument '--downsample' cannot be used with '--parameters <PARAMETERS>' or '--custom <PARAMETER_FILE>'");
    }
    if positional.len() < 2 {
        usage_error("The following required arguments were not provided:\n    <INPUT_FILE>\n    <OUTPUT_FILE>");
    }
    if positional.len() > 2 {
        usage_error(&format!("Found argument '{}' which wasn't expected, or isn't valid in this context", positional[2]));
    }
    o.input = positional[0].clone();
    o.output = positional[1].clone();
    o
}

fn decode_or_die(blob: &[u8]) -> Vec<f32> {
    srhip::rsr_decode(blob).unwrap_or_else(|_| die("ByteVec conversion failed"))
}

fn main() {
    let o = parse_args();

    // which graph, which parameters -- and the line the reference prints for each choice
    let (graph, params, banner): (i32, Vec<f32>, &str) = if let Some(ref file) = o.custom {
        let mut data = Vec::new();
        File::open(Path::new(file)).and_then(|mut f| f.read_to_end(&mut data)).unwrap_or_else(|_| die("Error opening parameter file"));
        (srhip::SR_GRAPH_SR_NET, decode_or_die(&data), "Upscaling using custom neural net parameters...")
    } else if o.downsample {
        (srhip::SR_GRAPH_DOWNSAMPLE, Vec::new(), "Downsampling using average pooling of linear RGB values...")
    } else {
        match o.parameters.as_ref().map(|s| s.as_str()).unwrap_or("imagenet") {
            "imagenetlinear" => (srhip::SR_GRAPH_SR_NET, decode_or_die(IMAGENETLINEAR), "Upscaling using linear loss imagenet neural net parameters..."),
            "anime" => (srhip::SR_GRAPH_SR_NET, decode_or_die(ANIME), "Upscaling using anime neural net parameters..."),
            "bilinear" => (srhip::SR_GRAPH_BILINEAR, Vec::new(), "Upscaling using bilinear interpolation..."),
            _ => (srhip::SR_GRAPH_SR_NET, decode_or_die(IMAGENET), "Upscaling using imagenet neural net parameters..."),
        }
    };
    print!("{}", banner);
    let _ = stdout().flush();

    // a wrong parameter count comes back as SR_E_PARAM_COUNT, whose text is the reference's assert message
    let mut engine = Engine::new(graph, &params, o.device).unwrap_or_else(|e| die(&e));
    if graph == srhip::SR_GRAPH_SR_NET && o.split_f16 {
        engine.set_precision(srhip::SR_PRECISION_SPLIT_F16).unwrap_or_else(|e| die(&e));
    }

    let rgba = image::open(Path::new(&o.input)).unwrap_or_else(|_| die("Error opening input image file.")).to_rgba();
    let (w, h) = rgba.dimensions();
    let out = engine.upscale_rgba8(&rgba.into_raw(), w, h).unwrap_or_else(|e| die(&e));
    if o.timing {
        let (total, h2d, d2h) = engine.last_timing();
        let _ = writeln!(std::io::stderr(), "\n[timing] kernels {:.3} ms, h2d {:.3} ms, d2h {:.3} ms", total, h2d, d2h);
    }

    print!(" Writing file...");
    let _ = stdout().flush();
    let (ow, oh) = engine.out_dims(w, h);
    let img = image::RgbaImage::from_raw(ow, oh, out).expect("engine returned a full RGBA8 buffer");
    img.save(Path::new(&o.output)).unwrap_or_else(|_| die("Could not write output file"));
    println!(" Done");
}
