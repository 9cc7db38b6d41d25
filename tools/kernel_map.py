"""Stage sums of the kernel map at the top of DESIGN.md from the committed serial step summaries:
    python tools/kernel_map.py [profiles/round6_serial_step_summary.txt profiles/round6_config3_serial_step_summary.txt]
prints one line per stage: launches and microseconds per pass (fp32 | bf16).  Kernel -> stage by name prefix."""
import re
import sys

STAGES = [
    ("sweep assembly", ("sweep_count", "sweep_scan", "sweep_write")),
    ("voxelizer + mean reader", ("vox_",)),
    ("index pyramid + rulebooks", ("idx_", "rows_place", "rows_permute", "rulebook_kernel", "block_work_kernel", "range_split_kernel")),
    ("sparse conv 128->128", ("spconv_f32_compact<128, 128", "spconv_bf16_win<128, 128", "spconv_bf16_ws<128, 128")),
    ("sparse conv 64->64", ("spconv_f32_compact<64, 64", "spconv_bf16_win<64, 64")),
    ("sparse conv 32->32", ("spconv_f32_c32<32", "spconv_bf16_ws<32, 32")),
    ("strided convs", ("spconv_f32_compact<16, 32", "spconv_f32_compact<32, 64", "spconv_f32_compact<64, 128", "spconv_bf16_ws<16, 32", "spconv_bf16_ws<32, 64",
                       "spconv_bf16_ws<64, 128")),
    ("sparse conv 16->16", ("spconv_f32_res16", "spconv_bf16_ws<16, 16")),
    ("densify", ("densify_",)),
    ("RPN + CenterHead convs", ("conv2d_", "conv1x1_")),
    ("decode + rotated NMS", ("dec_", "nms_")),
    ("forecast association", ("det_to_global", "forecast_")),
]


def read(path):
    rows = []
    for line in open(path):
        m = re.match(r"\s*([\d.]+) us\s+(\d+) x\s+(.*)", line)
        if m:
            rows.append((float(m.group(1)), int(m.group(2)), m.group(3)))
    return rows


def main():
    paths = sys.argv[1:] or ["profiles/round6_serial_step_summary.txt", "profiles/round6_config3_serial_step_summary.txt"]
    tables = [read(p) for p in paths]
    for name, prefixes in STAGES:
        cells = []
        for rows in tables:
            hit = [(us, n) for us, n, k in rows if any(k.startswith(p) for p in prefixes)]
            cells.append("%3d launches %7.1f us" % (sum(n for _, n in hit), sum(us for us, _ in hit)))
        print("%-28s %s" % (name, " | ".join(cells)))
    for p, rows in zip(paths, tables):
        known = sum(us for us, n, k in rows if any(k.startswith(pf) for _, pfs in STAGES for pf in pfs))
        print("%s: %.1f us in the stages above, %.1f us elsewhere (copies, fills)" % (p, known, sum(us for us, _, _ in rows) - known))


if __name__ == "__main__":
    main()
