"""
pipe_mix.py - GPU-BOX TOOLING: do FP64 MFMA and vector work of two waves on one SIMD overlap?
(libqocx_diag.so, knob "peak_mode": qocx_kernels.hip pipe_mix_kernel.) Every wave issues the same
number of cycles alone - 8 MFMAs of 64 cycles or 128 vector FMAs of 4 - so with two waves per SIMD a
mixed launch takes as long as a one-wave launch if the pipes are independent and as long as a
two-wave launch of one kind if they are shared.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from tools import diaglib
    diaglib.load()
    from qoc_amd.engine import Engine
    engine = Engine(0)
    iters = 20000
    out = {}
    for waves in (1, 2):
        for mode, name in ((0, "mfma"), (1, "f64 vector"), (2, "mfma | f64 vector"), (3, "f32 vector"),
                           (4, "mfma | f32 vector")):
            engine.set_knob("peak_mode", mode)
            tf = engine.mfma_peak(waves, iters)
            # the entry point reports blocks * iters * 8 * 2048 flop / time: recover the time
            blocks = 256 * 4 * waves
            ms = blocks * iters * 8.0 * 2048.0 / (tf * 1e12) * 1e3
            out["%d wave(s)/SIMD, %s" % (waves, name)] = round(ms, 3)
    print(json.dumps(out, indent=1))
    engine.close()


if __name__ == "__main__":
    main()
