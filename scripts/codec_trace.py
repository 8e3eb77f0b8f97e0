"""One warm DAC encode + decode at B=8 under rocprofv3 --kernel-trace; scripts/codec_trace_report.py lists the dispatches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampnet_amd import synth as D
from vampnet_amd.codec import DacCodec
from vampnet_amd.engine import Engine
eng = Engine("cuda:0")
cfg = D.DAC_DEFAULT_CFG
codec = DacCodec(D.synth_dac_state_dict(cfg, 0), cfg, engine=eng, precision=os.environ.get("VN_CODEC_PRECISION", "bf16x3"))
audio = 0.1 * torch.randn(8, 1, 575 * 768, device="cuda")
codes = codec.encode(audio)["codes"]
codec.decode_codes(codes)
torch.cuda.synchronize()
marker = torch.zeros(7, device="cuda"); marker += 1          # marks the start of the measured pass in the trace
torch.cuda.synchronize()
codes = codec.encode(audio)["codes"]
torch.cuda.synchronize()
marker += 1
codec.decode_codes(codes)
torch.cuda.synchronize()
