"""Command-line demo: image file in -> upscaled image file out, through the runner C API.

The counterpart of the reference's only executable (realesrgan/mtk/cpp/jni/src/main.cpp:16-114 and
the edsr/rcan twins): same argument order, same init -> inference_ex -> save -> release sequence,
same report (sizes, scale, time, FPS) and the same return codes (0 ok, -1 on any failure).

  python -m superresolution_amd.demo <model_path> <input_image> <output_image> [--model realesrgan|edsr|rcan]

<model_path> is a weight file written by superresolution_amd.export (convert_checkpoint /
save_weights); the model family is taken from --model or, like the reference's file-name
heuristics (realesrgan_model.py:126-139), from the file name.  Image decoding / encoding uses
PIL in place of the reference's stb_image.
"""
import ctypes as C
import sys
import time

import numpy as np

from . import _lib


class RealesrganContext(C.Structure):  # include/realesrgan.h
    _fields_ = [("neuron_executor", C.c_void_p), ("input_width", C.c_int), ("input_height", C.c_int),
                ("input_channels", C.c_int), ("output_width", C.c_int), ("output_height", C.c_int),
                ("output_channels", C.c_int), ("scale_factor", C.c_int), ("input_size", C.c_size_t),
                ("output_size", C.c_size_t), ("initialized", C.c_int)]


class MeanShiftContext(C.Structure):   # include/edsr.h, include/rcan.h
    _fields_ = [("neuron_executor", C.c_void_p), ("input_width", C.c_int), ("input_height", C.c_int),
                ("input_channels", C.c_int), ("output_width", C.c_int), ("output_height", C.c_int),
                ("output_channels", C.c_int), ("scale_factor", C.c_int), ("rgb_mean", C.c_float * 3),
                ("rgb_range", C.c_float), ("input_size", C.c_size_t), ("output_size", C.c_size_t),
                ("initialized", C.c_int)]


TITLES = {"realesrgan": "Real-ESRGAN", "edsr": "EDSR", "rcan": "RCAN"}


def family_from_name(path):
    name = path.lower()
    for fam in ("edsr", "rcan"):
        if fam in name:
            return fam
    return "realesrgan"


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    family = None
    if "--model" in argv:
        i = argv.index("--model")
        family = argv[i + 1] if i + 1 < len(argv) else None
        del argv[i:i + 2]
        if family not in TITLES:
            print(f"[ERROR] --model must be one of {sorted(TITLES)}", file=sys.stderr)
            return -1
    if len(argv) != 4:
        print(f"Usage: {argv[0]} <model_path> <input_image> <output_image> [--model realesrgan|edsr|rcan]")
        print(f"Example: {argv[0]} RealESRGAN_x4plus.srmi input.png output.png")
        return -1
    model_path, input_path, output_path = argv[1:4]
    family = family or family_from_name(model_path)
    title = TITLES[family]

    print("========================================")
    print(f"{title} Super-Resolution (MI355X)")
    print("========================================")
    print(f"Model:   {model_path}")
    print(f"Input:   {input_path}")
    print(f"Output:  {output_path}")
    print("========================================")

    L = _lib.lib()
    ctx = RealesrganContext() if family == "realesrgan" else MeanShiftContext()
    init = getattr(L, f"{family}_init")
    infer_ex = getattr(L, f"{family}_inference_ex")
    release = getattr(L, f"{family}_release")

    t0 = time.perf_counter()
    ret = init(model_path.encode(), C.byref(ctx))
    if ret != 0:
        print(f"Failed to initialize {title} model!", file=sys.stderr)
        return -1
    print(f"[INFO] Initialization time: {(time.perf_counter() - t0) * 1e3:.3f} ms")

    print("\n[INFO] Loading input image...")
    try:
        from PIL import Image
        with Image.open(input_path) as im:
            channels = len(im.getbands())
            img = np.ascontiguousarray(np.asarray(im.convert("RGB"), dtype=np.uint8))
    except Exception as e:  # unreadable file, unknown format
        print(f"[ERROR] Failed to load input image: {input_path} ({e})", file=sys.stderr)
        release(C.byref(ctx))
        return -1
    height, width = img.shape[:2]
    print(f"[INFO]   Original size: {width}x{height}x{channels}")

    print("\n[INFO] Running super-resolution...")
    out, ow, oh = C.c_void_p(), C.c_int(), C.c_int()
    t0 = time.perf_counter()
    ret = infer_ex(C.byref(ctx), img.ctypes.data_as(C.c_void_p), width, height, C.byref(out), C.byref(ow), C.byref(oh))
    ms = (time.perf_counter() - t0) * 1e3
    if ret != 0:
        print("[ERROR] Inference failed!", file=sys.stderr)
        release(C.byref(ctx))
        return -1

    print("\n========================================")
    print("Results:")
    print(f"  Input:    {width}x{height}")
    print(f"  Output:   {ow.value}x{oh.value}")
    print(f"  Scale:    x{ctx.scale_factor}")
    print(f"  Time:     {ms:.3f} ms")
    print(f"  FPS:      {1000.0 / ms:.3f}")
    print("========================================")

    print("\n[INFO] Saving output image...")
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    result = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_ubyte)), shape=(oh.value, ow.value, 3)).copy()
    libc.free(out)  # the library mallocs the output, the caller frees it (realesrgan.cpp:247, main.cpp:109)
    try:
        Image.fromarray(result, "RGB").save(output_path)
    except Exception as e:
        print(f"[ERROR] Failed to save output image: {output_path} ({e})", file=sys.stderr)
        release(C.byref(ctx))
        return -1
    print(f"[INFO] Output saved to: {output_path}")
    release(C.byref(ctx))
    print("\n[INFO] Done!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
