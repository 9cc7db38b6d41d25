"""The op_sel packed-fp32 victim (tools/probes/alu_probe.hip modes 19 / 21, and the exact forms 16 / 14 as controls) next to SYNTHETIC disturbers:
waves that issue one kind of instruction back to back -- v_mfma_f32_32x32x16_bf16, v_mfma_f32_16x16x32_bf16, the fp32 MFMAs, plain v_fma_f32.
    python tools/soak_alu_synth.py [rounds]"""
import ctypes
import os
import sys

import torch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "libalu_probe.so"))
P.alu_probe_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
P.alu_disturb_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
sa = torch.cuda.Stream()
sbs = [torch.cuda.Stream() for _ in range(3)]
mism = torch.zeros(1, dtype=torch.int64, device="cuda")
sink = torch.zeros(4, device="cuda")
victims = [(19, "pk_mul op_sel:[0,1]"), (21, "pk_add op_sel:[0,1]"), (16, "pk_mul plain"), (14, "pk_add plain")]
for kind, dname in [(-1, "nothing"), (0, "v_mfma_f32_32x32x16_bf16"), (1, "v_mfma_f32_16x16x32_bf16"), (2, "fp32 MFMAs"), (3, "v_fma_f32 chains")]:
    line = "%-26s" % dname
    for mode, nm in victims:
        mism.zero_()
        torch.cuda.synchronize()
        for r in range(rounds):
            if kind >= 0:
                for sb in sbs:  # 512 workgroups of four waves, ~1 ms each: two rounds over the chip, busy for the victim's whole run
                    P.alu_disturb_run(kind, 512, 20000, sink.data_ptr(), sb.cuda_stream)
            for _ in range(6):
                P.alu_probe_run(mode, 128, 24, mism.data_ptr(), sink.data_ptr(), sa.cuda_stream)
            torch.cuda.synchronize()
        line += " | %s: %d" % (nm, int(mism.item()))
    print(line, flush=True)
