"""Regenerates the measured table of DESIGN.md section 5 (between the BENCH_TABLE markers) from profiles/r06_bench_lines.json,
the bench lines of the round's evidence run (tools/final_round.sh)."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_lines.json")).read().strip().split("\n")[-1])
c, rl, cb = h["config"], h["roofline"], h["cpu_baseline"]
ps = cb["parity_sampled"]
rows = [
    "| workload (bench leg) | round 5 (driver record `BENCH_r05.json`) | round 6 (`profiles/r06_bench_lines.json`) | time per pass | roofline of the leg | CPU oracle on the box | parity in the same run |",
    "|---|---|---|---|---|---|---|",
    "| **headline** 108-atom LJ(1,1), 16 384 replicas x 49 steps + RDF + adjoint (configs #1/#2) | 34.32 M steps/s | **%.2f M steps/s** | %.2f ms | VALU: %.3f of the 157.3 TF data-sheet peak useful, executed %.3f; executed = %.2f of the measured 118 TF; `valu_busy` %.2f; HBM: %.3f of 8 TB/s measured, %.3f compulsory (frames in, costates out) | %.0f steps/s on %d threads | replicas 0 / 8 192 / 16 383 of the timed launch vs oracle: max \\|dq\\| %.1e, \\|dg\\| %.1e, dtheta %.1e |"
    % (h["value"] / 1e6, h["ms_per_step"], rl["frac"], rl["executed_frac"], rl["executed_frac_of_measured_peak"], rl["valu_busy"],
       rl["hbm_frac_measured"], rl["hbm_frac_compulsory"], cb["value"], cb["cores"], ps["max_abs_dq"], ps["max_abs_dg"], ps["rel_dtheta"]),
    "| `exvol108`: the README's ExcludedVolume(sigma 1, eps 1, p 12) at dt 0.01 | 34.49 M | %.2f M steps/s | %.2f ms | the same ring kernels (attractive coefficient 0): %.3f | %.0f | \\|dq\\| %.1e |"
    % (c["exvol108_md_steps_per_s"] / 1e6, c["exvol108_ms_per_pass"], c["exvol108_kernel_frac"], c["exvol108_cpu_steps_per_s"],
       c["exvol108_parity_max_abs_dq"]),
    "| `schnet4096`: 8 stacked 4 096-bead systems, A64 F128 G30 2 conv + prior, bf16 filter operands + bf16 gathered node rows (round 6), 52-step passes (config #5, SURVEY 8d M4) | 3 323 (bf16 operands, f32 rows) | **%.0f steps/s**; bf16 operands with f32 rows %.0f; all-f32 %.0f; 10-step passes %.0f | %.1f ms | step vs mixed MFMA roof %.3f (f32: %.3f); dominant kernel `cfconv_bwd_bf16<...,true,true,true>` %.3f of the bf16 roof | %.2f (one 4 096-bead replica, 16 threads) | replicas 0 / 7 of the timed stack vs oracle, 2 steps: \\|dq\\| %.1e A, dtheta %.1e; 52 steps vs the all-f32 path: \\|dq\\| %.1e A, dtheta %.1e of the largest entry |"
    % (c["schnet4096_md_steps_per_s"], c["schnet4096_bf16_f32rows_md_steps_per_s"], c["schnet4096_f32_md_steps_per_s"],
       c["schnet4096_10step_md_steps_per_s"], c["schnet4096_ms_per_pass"], c["schnet4096_step_roof_frac"],
       c["schnet4096_f32_step_roof_frac"], c["schnet4096_kernel_frac"], c["schnet4096_cpu_steps_per_s"],
       c["schnet4096_parity_max_abs_dq"], c["schnet4096_parity_rel_dtheta"], c["schnet4096_52step_vs_f32_max_abs_dq"],
       c["schnet4096_52step_vs_f32_rel_dtheta"]),
    "| `lj4096`: 64 x 4 096-atom LJ liquid, 50 steps + RDF every 5th frame + adjoint (config #4) | 188.7 k | **%.1f k steps/s** | %.2f ms | B_step over the step time: %.3f of HBM -- the pass is VALU-issue bound (section 5) | %.2f | last replica of the timed launch, 2 steps: \\|dq\\| %.1e, dtheta %.1e; 16 steps with device-side rebuilds: `tests/test_gpu_secondary_pins.py` |"
    % (c["lj4096_md_steps_per_s"] / 1e3, c["lj4096_ms_per_pass"], c["lj4096_kernel_frac"], c["lj4096_cpu_steps_per_s"],
       c["lj4096_parity_max_abs_dq"], c["lj4096_parity_rel_dtheta"]),
    "| `water192`: config #3, SchNet A128 F128 G32 3 conv + prior, one system, f32, graph replay | 898 | **%.0f steps/s** | %.1f ms (20 steps) | %.3f of f32 MFMA (≈ 100 graph nodes per step, 94 %% GPU-busy: kernels of one round of workgroups each) | %.1f | vs the REFERENCE's own run (golden G14): \\|dq\\| %.1e A, dtheta %.1e |"
    % (c["water192_md_steps_per_s"], c["water192_ms_per_pass"], c["water192_kernel_frac"], c["water192_cpu_steps_per_s"],
       c["water192_parity_max_abs_dq"], c["water192_parity_rel_dtheta"]),
    "| `water192x64` (round 6): 64 copies of config #3's box stacked in ONE trajectory (12 288 atoms, 325 k edges), f32 | -- | **%.0f steps/s** (%.1f x one system) | %.1f ms (64 x 20 steps) | %.3f of f32 MFMA | -- | replicas 0 and 63 (all 64) on the REFERENCE's own run (G14): \\|dq\\| %.1e A, dtheta %.1e |"
    % (c["water192x64_md_steps_per_s"], c["water192x64_md_steps_per_s"] / c["water192_md_steps_per_s"], c["water192x64_ms_per_pass"],
       c["water192x64_kernel_frac"], c["water192x64_parity_max_abs_dq"], c["water192x64_parity_rel_dtheta"]),
    "| one 4 096-bead SchNet system per GPU (config #5 as written) | 1 249 | %.0f steps/s | %.0f us per step | ~70 graph nodes per step, 90 %% GPU-busy: 5–35 µs kernels of one round of workgroups each | -- | pinned in `tests/test_gpu_secondary_pins.py` |"
    % (c["single_system_md_steps_per_s"], c["single_system_us_per_md_step"]),
]
table = "\n".join(rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
new = re.sub(r"<!-- BENCH_TABLE_BEGIN -->.*?<!-- BENCH_TABLE_END -->", "<!-- BENCH_TABLE_BEGIN -->\n" + table.replace("\\", "\\\\") + "\n<!-- BENCH_TABLE_END -->", s, flags=re.S)
assert new != s or table in s, "markers not found"
open(path, "w").write(new)
print(table)
