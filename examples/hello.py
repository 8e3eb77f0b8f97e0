"""The reference's hello.py (encode -> build_mask -> vamp -> decode -> write) on the MI355X engine.

With real checkpoints:   python examples/hello.py --coarse coarse.pth --c2f c2f.pth --codec codec.pth --input in.wav
Without (this image has no network / no HF hub): seeded random-init weights of the real architecture and a synthetic
input signal — the output is noise-like audio, but every stage of the pipeline runs at full size and is timed.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import Interface                                    # noqa: E402
from vampnet_amd import synth as W                                   # noqa: E402
from vampnet_amd.codec import AudioSignal, DacCodec                  # noqa: E402
from vampnet_amd.synth import model_kwargs                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--coarse"), ap.add_argument("--c2f"), ap.add_argument("--codec"), ap.add_argument("--input")
ap.add_argument("--output", default="gpurun_out/hello_output.wav")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--rng", choices=["device", "torch_device", "torch"], default="torch_device",
                help="device: Philox noise on the GPU; torch_device: torch's CPU mt19937 stream continued on the GPU, so seeded "
                     "runs are token-identical to the reference at device speed; torch: the same numbers drawn on the host "
                     "(9.4 MB of Exp(1) noise per sampling step, ~0.6 s per clip)")
args = ap.parse_args()

t0 = time.perf_counter()
if args.coarse and args.c2f and args.codec:
    interface = Interface(coarse_ckpt=args.coarse, coarse2fine_ckpt=args.c2f, codec_ckpt=args.codec, device="cuda:0",
                          max_batch=args.batch, rng=args.rng)
else:
    print("no checkpoints given: seeded random-init weights of the real architecture")
    codec_sd = W.synth_dac_state_dict(W.DAC_DEFAULT_CFG, 0)
    codec = DacCodec(codec_sd, W.DAC_DEFAULT_CFG, device="cuda:0")
    interface = Interface.from_state_dicts(codec, W.synth_state_dict(W.COARSE_DIMS, 0), model_kwargs(W.COARSE_DIMS),
                                           W.synth_state_dict(W.C2F_DIMS, 1), model_kwargs(W.C2F_DIMS), device="cuda:0",
                                           max_batch=args.batch, rng=args.rng)
print(f"models ready in {time.perf_counter() - t0:.1f} s")

if args.input:
    signal = AudioSignal.from_wav(args.input)
else:
    t = np.arange(10 * 44100) / 44100.0
    signal = AudioSignal((0.2 * np.sin(2 * np.pi * 220 * t) + 0.1 * np.sin(2 * np.pi * 3000 * t)).astype(np.float32), 44100)


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"{name:12s} {1e3 * (time.perf_counter() - t):8.1f} ms")
    return out


for it in range(2):                        # second pass = warm timings
    print("cold pass" if it == 0 else "warm pass")
    codes = timed("encode", lambda: interface.encode(signal))                                   # (1, 14, T)
    mask = timed("build_mask", lambda: interface.build_mask(codes, signal, periodic_prompt=13, upper_codebook_mask=3))
    out = timed("vamp", lambda: interface.vamp(codes, mask, batch_size=args.batch, return_mask=False, temperature=1.0,
                                                typical_filtering=False, seed=0))
    sig = timed("decode", lambda: interface.decode(out))
os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
sig.cpu().write(args.output)
print("tokens", tuple(out.shape), "-> audio", tuple(sig.samples.shape), "written to", args.output)
