"""DAC encode / decode throughput on the engine (rows a18/a19; parity unpinned).  10 s @ 44.1 kHz clips, default config."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import synth as D             # seeded synthetic codec weights / configs (data generators only)
from vampnet_amd.codec import DacCodec
from vampnet_amd.engine import Engine

eng = Engine("cuda:0")
cfg = D.DAC_DEFAULT_CFG
for precision, B in (("f16x2", 1), ("f16x2", 8), ("bf16x3", 8), ("f32", 8)):
    codec = DacCodec(D.synth_dac_state_dict(cfg, 0), cfg, engine=eng, precision=precision)
    audio = 0.1 * torch.randn(B, 1, 575 * 768, device="cuda")
    for name, fn in (("encode", lambda: codec.encode(audio)["codes"]), ("decode", None)):
        if name == "decode":
            codes = codec.encode(audio)["codes"]
            fn = lambda: codec.decode_codes(codes)
        for _ in range(2): fn()
        torch.cuda.synchronize()
        eng.profile_begin(4000)
        t0 = time.perf_counter()
        n = 3
        for _ in range(n): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        st = eng.profile_end()["conv1d"]
        print(json.dumps({"op": name, "pipe": precision, "batch": B, "ms": round(dt * 1e3, 2), "clip_seconds_per_s": round(B * 10.0 / dt, 1),
                          "conv_launches": int(st[0] / n), "conv_ms": round(st[1] / n, 2), "conv_TF": round(st[2] / st[1] / 1e9, 1),
                          "conv_GFLOP": round(st[2] / n / 1e9, 1), "conv_algorithmic_GB": round(st[3] / n / 1e9, 2)}), flush=True)
