#!/usr/bin/env python3
"""Where the attention kernel's key-tile iteration spends its cycles (diagnostic build with s_memtime stamps),
and the clock the chip holds meanwhile (s_memtime / s_memrealtime).  Run on the GPU box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip  # noqa: E402


def main():
    lib = hip.lib()
    res = {}
    for name, b, n, h in (("enc", 16, 1370, 16), ("info", 8, 2738, 12)):
        qkv = torch.randn(b * n, 3 * h * 64, device="cuda").bfloat16()
        qkv[:, : h * 64] = (qkv[:, : h * 64].float() * (0.125 * 1.4426950408889634)).bfloat16()
        out = torch.empty(b * n, h * 64, device="cuda", dtype=torch.bfloat16)
        for waves in (4, 2):
            nqb = (n + waves * 64 - 1) // (waves * 64)
            nwg = nqb * h * b
            diag = torch.zeros(nwg * 16, device="cuda", dtype=torch.int64)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(20):  # let the clock settle under load
                hip._check(lib.ufm_debug_attention_stamps(qkv.data_ptr(), out.data_ptr(), b, n, h, waves, diag.data_ptr(), st), "ufm_debug_attention_stamps")
            torch.cuda.synchronize()
            d = diag.view(nwg, 16).double().cpu()
            d = d[d[:, 5] > 0]  # persistent kernel: min(units, CUs x workgroups per CU) workgroups write a row
            tiles = d[:, 7]
            med = lambda x: float(x.median())  # noqa: E731
            row = {
                "per_tile_cycles": {k: med(d[:, i] / tiles) for i, k in enumerate(["sync_dma", "slot0", "slot1", "slot2", "slot3"])},
                "loop_cycles_per_tile": med(d[:, :5].sum(1) / tiles),
                "kernel_cycles_per_wg": med(d[:, 5]),
                "outside_loop_cycles_per_unit": med((d[:, 5] - d[:, :5].sum(1)) / d[:, 8]),
                "per_unit_cycles": {k: med(d[:, 9 + i] / d[:, 8]) for i, k in enumerate(["state_init", "prologue_wait_kreads", "seam_issue", "drain", "epilogue_compute", "store_issue"])},
                "clock_ghz": med(d[:, 5] / d[:, 6] * 0.1),
                "units": nwg, "workgroups": int(d.shape[0]),
            }
            res[f"{name}_w{waves}"] = row
            print(name, "waves", waves, json.dumps(row), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
