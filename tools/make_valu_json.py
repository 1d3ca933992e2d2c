#!/usr/bin/env python3
"""Builds profiles/valu.json and profiles/hbm_traffic.json from the committed rocprofv3 summaries of a round
(tools/final_profiles.sh <prefix> on the GPU box, summaries copied to profiles/).

usage: tools/make_valu_json.py <prefix, e.g. r02_d>

What bench.py's `roofline.valu` reports, per kernel / workload, every figure recomputable from the named files:
  wave_instr_per_sample   SQ_INSTS_VALU / camera samples of the launch
  lane_occupancy          SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  ns_per_valu_per_simd    kernel time * 1024 SIMDs / SQ_INSTS_VALU
  issue_frac_2clk         against one wave64 VALU instruction per 2 clocks per SIMD at 2.4 GHz (the guide's figure)
  issue_frac_ubench       against what profiles/<prefix>_valu_peak_ubench.txt measures at 6 waves per SIMD for this kernel's mix of plain
                          and quarter-rate instructions (the ceiling this GPU actually reaches)
  mix_model_ns,           what profiles/<prefix>_salu_mix_ubench.txt measures for a VALU stream with this kernel's SALU instructions per VALU
  issue_frac_mix_model    instruction (interpolated), plus the quarter-rate surcharge of its VALU mix; and that figure over the kernel's
  wait_frac / stall_frac  SQ_WAIT_ANY, SQ_WAIT_INST_ANY over SQ_WAVE_CYCLES
  spill                   scratch bytes per lane of the dispatch, scratch loads / stores per launch, WRITE_SIZE
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
SAMPLES = {"cornell": 1024 * 768 * 1024, "veach": 1280 * 720 * 1024, "light_mis": 1024 * 768 * 1024, "generic": 1024 * 768 * 1024, "recursion": 1024 * 768 * 1024,
           "stress": 4096 * 4096 * 256,                       # configs[4]'s geometry (4096 x 4096, depth 16) at 256 of its 16 384 spp: the kernel's rate does not depend on spp
           "single": 1024 * 1024 * 512,                       # ky's default driver (Cornell box under the environment light, 1024 x 1024) at 512 spp
           "batch": 1024 * 1024 * (5 * 256 + 1)}              # configs[3]'s six frames at 256 of their 2048 spp (five path frames + the 1-spp AOV pass)
LABEL = {"cornell": "render_kernel<strategy 48, feat 3975 (one rectangle area light that is its own carrier, small tables, walls and lamp housing as boxes, every planar surface in an axis plane, plastic on rectangles only), integrator 11> on BASELINE configs[1] (Cornell 1024x768x1024)",
         "veach": "render_kernel<strategy 48, deferred shadow rays, feat 6372 (sphere lights, no delta lobes, small tables, plastic on rectangles only, tilted rectangles are planks about the x axis)> on configs[2]'s scene at 1024 spp (Veach 1280x720)",
         "light_mis": "render_kernel<strategy 32> (the light_mis instantiation) on configs[1]'s scene",
         "recursion": "render_kernel<strategy 48, feat 263, integrator 9> (path_tracing_recursion_t) on configs[1]'s scene",
         "generic": "render_kernel<false,-1> (strategy read at run time; KYHIP_SPECIALISE=0) on configs[1]'s scene with direct_sample light_mis",
         "stress": "render_kernel<strategy 48, feat 3975> on configs[4]'s geometry (Cornell 4096x4096, depth 16) at 256 spp",
         "single": "render_kernel<strategy 48, feat 3728 (one environment light, small tables, walls as a box, axis planes only, plastic on rectangles only), integrator 11> on ky's render_single_scene frame at 512 spp",
         "batch": "the six launches of configs[3]'s step (four Cornell light variants, Veach square, AOV pass; 1024x1024) at 256 spp: counters summed over its kernels"}


def counters(path):
    """-> ({counter: value per launch SET}, kernel ms per launch set, scratch bytes of the first render_kernel dispatch).
    A launch set = the render_kernel dispatches of one step of the workload: one for the single-frame workloads, six for the batch (whose frames run on
    several instantiations: their lines are summed).  The summary lists, per kernel name, the dispatch count and the average per dispatch."""
    sums, disp, ms_total, calls_total, scratch = {}, {}, 0.0, 0, None
    for line in open(path):
        m = re.match(r"(.*render_kernel.*?)\s(SQ_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([0-9.]+)\s*$", line)
        if m:
            sums[m.group(2)] = sums.get(m.group(2), 0.0) + float(m.group(4)) * int(m.group(3))
            disp.setdefault(m.group(2), 0)
            disp[m.group(2)] += int(m.group(3))
        m = re.match(r"void render_kernel.*?\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m:
            calls_total += int(m.group(1))
            ms_total += float(m.group(2)) / 1e3
        m = re.search(r"render_kernel dispatch: .*scratch=(\d+)", line)
        if m and scratch is None:
            scratch = int(m.group(1))
    per_set = FRAMES_PER_SET.get(CURRENT[0], 1)
    out = {k: v * per_set / disp[k] for k, v in sums.items()}
    ms = (ms_total * per_set / calls_total) if calls_total else None
    return out, ms, scratch


CURRENT = ["cornell"]      # the workload whose files are being read (counters() needs its launches per set)
FRAMES_PER_SET = {"batch": 6}


def ubench(path):
    """ns per wave-instruction per SIMD at 6 waves/SIMD for the instruction classes of valu_peak.hip"""
    res = {}
    for line in open(path):
        m = re.match(r"(.+?)\s+waves/SIMD 6:.*\(([0-9.]+) ns\)", line)
        if m:
            res[m.group(1).strip()] = float(m.group(2))
    return res


def salu_curve(path):
    """ns per v_fmac slot at 6 waves/SIMD against SALU instructions per VALU instruction (tools/ubench/salu_mix.hip)"""
    pts = {}
    ratio = {"v_fmac alone": 0.0, "v_fmac + 1/8 SALU": 0.125, "v_fmac + 1/4 SALU": 0.25, "v_fmac + 1/2 SALU": 0.5, "v_fmac + 3/4 SALU": 0.75, "v_fmac + 1 SALU": 1.0}
    for line in open(path):
        m = re.match(r"(.+?)\s+waves/SIMD 6:.*?([0-9.]+) ns per slot", line)
        if m and m.group(1).strip() in ratio:
            pts[ratio[m.group(1).strip()]] = float(m.group(2))
    return sorted(pts.items())


def interpolate(curve, x):
    for (x0, y0), (x1, y1) in zip(curve, curve[1:]):
        if x0 <= x <= x1:
            return y0 + (y1 - y0) * (x - x0) / (x1 - x0)
    return curve[-1][1]


def main():
    prefix = sys.argv[1]
    ub = ubench(os.path.join(PROF, prefix + "_valu_peak_ubench.txt"))
    curve = salu_curve(os.path.join(PROF, prefix + "_salu_mix_ubench.txt"))
    plain, trans = ub["v_fmac_f32_e32 (VOP2, 3 vgpr)"], ub["v_rcp_f32"]
    valu, traffic = {}, None
    # the kernels the counters were measured on (tools/final_profiles.sh writes kyhip_kernel_source_hash(): device headers + compile flags): bench.py
    # withholds these figures when it runs a build whose kernels come from other sources
    try:
        lib_sha = open(os.path.join(PROF, prefix + "_kernel_source_hash.txt")).read().split()[0]
    except Exception:
        lib_sha = None
    for wl in ("cornell", "veach", "light_mis", "generic", "recursion", "stress", "batch", "single"):
        if not os.path.exists(os.path.join(PROF, "%s_%s_pmc_sq_issue.txt" % (prefix, wl))):
            continue
        CURRENT[0] = wl
        base = os.path.join(PROF, "%s_%s_" % (prefix, wl))
        issue, _, scratch = counters(base + "pmc_sq_issue.txt")
        mix, _, _ = counters(base + "pmc_sq_mix.txt")
        _, ms, _ = counters(base + "kernel_stats.txt")
        fetch, _, _ = counters(base + "hbm_fetch_size.txt")
        write, _, _ = counters(base + "hbm_write_size.txt")
        n = SAMPLES[wl]
        insts = issue["SQ_INSTS_VALU"]
        f_trans = mix["SQ_INSTS_VALU_TRANS_F32"] / insts
        ns = ms * 1e6 * 1024 / insts
        # the engine clock the launch really ran at: SQ_BUSY_CYCLES is summed over the shader engines' SQs (32 on this chip: the kernel keeps every one busy
        # from start to end), so busy cycles per SQ over the kernel's duration is the clock
        clock_ghz = issue["SQ_BUSY_CYCLES"] / 32 / (ms * 1e6) if "SQ_BUSY_CYCLES" in issue else None
        ceiling = (1 - f_trans) * plain + f_trans * trans
        fetch_b, write_b = fetch["FETCH_SIZE"] * 1024 * 2, write["WRITE_SIZE"] * 1024   # KiB; FETCH_SIZE x 2 per the guide's gfx950 note
        valu[wl] = {
            "kernel": LABEL[wl], "samples_per_launch": n, "kernel_ms": ms, "kernel_source_hash": lib_sha,
            "wave_instr_per_sample": insts / n,
            "lane_occupancy": mix["SQ_THREAD_CYCLES_VALU"] / (64 * issue["SQ_ACTIVE_INST_VALU"]),
            "ns_per_valu_per_simd": ns,
            "issue_frac_2clk": (2 / 2.4) / ns,
            "clock_ghz_from_sq_busy_cycles": clock_ghz, "issue_frac_2clk_at_measured_clock": ((2 / clock_ghz) / ns) if clock_ghz else None,
            "ubench_ceiling_ns": ceiling, "issue_frac_ubench": ceiling / ns,
            "quarter_rate_fraction": f_trans, "salu_per_valu": issue["SQ_INSTS_SALU"] / insts,
            # what a v_fmac stream with this much scalar company reaches, plus what the kernel's quarter-rate instructions add to a plain stream
            "mix_model_ns": interpolate(curve, issue["SQ_INSTS_SALU"] / insts) + (ceiling - plain),
            "issue_frac_mix_model": (interpolate(curve, issue["SQ_INSTS_SALU"] / insts) + (ceiling - plain)) / ns,
            "wait_frac": issue["SQ_WAIT_ANY"] / issue["SQ_WAVE_CYCLES"], "stall_frac": issue["SQ_WAIT_INST_ANY"] / issue["SQ_WAVE_CYCLES"],
            "spill": {"scratch_bytes_per_lane": scratch, "scratch_loads_per_launch": mix["SQ_INSTS_VMEM_RD"], "scratch_stores_per_launch": mix["SQ_INSTS_VMEM_WR"],
                      "write_size_bytes_per_launch": write_b},
            "hbm_bytes_per_launch": fetch_b + write_b,
            "files": [os.path.basename(base) + s for s in ("kernel_stats.txt", "pmc_sq_issue.txt", "pmc_sq_mix.txt", "hbm_fetch_size.txt", "hbm_write_size.txt")]
                     + [prefix + "_valu_peak_ubench.txt", prefix + "_salu_mix_ubench.txt"],
        }
        if wl == "cornell":
            traffic = {"_comment": "HBM traffic of ONE render_kernel launch of the bench workload from rocprofv3 --pmc (two separate passes; FETCH_SIZE / WRITE_SIZE are in "
                                   "KiB; FETCH_SIZE doubled per /opt/skills/guides/MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE as is).  The kernel keeps all path state in "
                                   "registers and LDS: what reaches the memory side is the write-back of spilled registers, the fixed-point pixel atomics and the scene.",
                       "workload": "cornell", "width": 1024, "height": 768, "spp": 1024, "samples_per_launch": n,
                       "fetch_size_kib": fetch["FETCH_SIZE"], "write_size_kib": write["WRITE_SIZE"], "hbm_bytes_per_launch": fetch_b + write_b, "round": prefix}
    with open(os.path.join(PROF, "valu.json"), "w") as fh:
        json.dump(valu, fh, indent=1)
    with open(os.path.join(PROF, "hbm_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    for wl, v in valu.items():
        print("%-8s %.1f VALU/sample  lanes %.3f  %.3f ns/instr/SIMD  issue %.2f of 2-clk, %.2f of ubench, %.2f of the VALU+SALU model (%.2f SALU/VALU)  wait %.2f stall %.2f  scratch %s B/lane  HBM %.2f GB/launch" % (
            wl, v["wave_instr_per_sample"], v["lane_occupancy"], v["ns_per_valu_per_simd"], v["issue_frac_2clk"], v["issue_frac_ubench"], v["issue_frac_mix_model"], v["salu_per_valu"], v["wait_frac"], v["stall_frac"],
            v["spill"]["scratch_bytes_per_lane"], v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
