"""Turn the PMC passes of tools/final_profiles.sh into profiles/roundN/traffic.json (read by bench.py):

    python tools/traffic_json.py gpurun_out/final profiles/round2/traffic.json [tag]

Per kernel: HBM bytes per unit of work from FETCH_SIZE / WRITE_SIZE (own --pmc passes, unit KB), corrected as
MI355X_MICROARCH.md prescribes for gfx950 -- FETCH_SIZE counts 16-B-per-lane coalesced reads at one half, which
k_fold_planes (a pure stream of such reads, known size) confirms in the same runs -- and how busy the vector issue port was:
SQ_INSTS_VALU x the mean issue cost of the kernel's inner-loop instruction mix (profiles/round2/isa_mix.json: 2 / 4 / 8 cycles per
wave64 instruction by class, measured by tools/micro/valu_issue.hip) over SIMD-cycles (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import json
import re
import sys

import os
src, dst = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "r3"
rnd = os.path.basename(os.path.dirname(os.path.abspath(dst)))
TWO_LEVEL_RUN = "final_tl" in src
mix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(dst)), "isa_mix.json")))["kernels"]
if TWO_LEVEL_RUN:
    mix = {k.replace(", two-level>", ">"): v for k, v in mix.items() if "two-level" in k or k.startswith("k_shade")}


def table(name):
    out = {}
    for line in open(f"{src}/pmc_{name}.txt"):
        # the production instantiations: k_shade one tile per workgroup (k_shade<false, false, true> is the tile-walking safety net), the
        # traversal kernels for scenes that are one world-space tree (<., true>: the ones that enter instances)
        for long, short in (("k_shade<false, false, false, false>", "k_shade<false>"), ("k_shade<false, false, false>", "k_shade<false>"), ("k_shade<false, false>", "k_shade<false>"),
                            ("k_trace<true, false, false>", "k_trace<true>"), ("k_trace<false, false, false>", "k_trace<false>"),  # round 4: a third parameter (DESCENT)
                            # round 6: <any hit, LEVELS> -- 0 one world-space tree, 1 folded / parked instances, 2 the general instance route
                            ("k_trace<true, 0>", "k_trace<true>"), ("k_trace<false, 0>", "k_trace<false>"),
                            ("k_trace<true, 1>", "k_trace<true>" if TWO_LEVEL_RUN else "k_trace<true,two-level>"),
                            ("k_trace<false, 1>", "k_trace<false>" if TWO_LEVEL_RUN else "k_trace<false,two-level>"),
                            ("k_trace<true, 2>", "k_trace<true>" if TWO_LEVEL_RUN else "k_trace<true,general>"),
                            ("k_trace<false, 2>", "k_trace<false>" if TWO_LEVEL_RUN else "k_trace<false,general>"),
                            ("k_trace<true, true, false>", "k_trace<true>" if TWO_LEVEL_RUN else "k_trace<true,two-level>"),
                            ("k_trace<false, true, false>", "k_trace<false>" if TWO_LEVEL_RUN else "k_trace<false,two-level>"),
                            ("k_trace<true, false>", "k_trace<true>"), ("k_trace<false, false>", "k_trace<false>"),
                            ("k_trace_packet<false, false>", "k_trace_packet<false>"),
                            ("k_trace_multi<4, false>", "k_trace_multi<4>"),
                            ("k_trace_multi<4, true>", "k_trace_multi<4>" if TWO_LEVEL_RUN else "k_trace_multi<4,two-level>"),
                            # passes taken with every instance entered (gpurun_out/final_tl): the <., true> instantiations ran; bench.py names
                            # the kernels of its line by what they compute either way
                            ("k_trace<true, true>", "k_trace<true>" if TWO_LEVEL_RUN else "k_trace<true,two-level>"),
                            ("k_trace<false, true>", "k_trace<false>" if TWO_LEVEL_RUN else "k_trace<false,two-level>"),
                            ("k_trace_packet<false, true>", "k_trace_packet<false>" if TWO_LEVEL_RUN else "k_trace_packet<false,two-level>")):
            line = line.replace(long, short)
        m = re.match(r"(?:void )?ptd::(\S+)\s+(\S+)\s+(\d+)\s+per-dispatch\s+(\d+)\s+\((\d+) dispatches\)", line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), int(m.group(5)))
    return out


bench = json.loads(open(f"{src}/bench.json").read().strip().splitlines()[-1])
kern = bench["roofline"]["kernels"]
fetch, write = table("FETCH_SIZE"), table("WRITE_SIZE")
insts, act = table("SQ_INSTS_VALU"), table("GRBM_GUI_ACTIVE")
try:
    waves = table("SQ_WAVES")
except OSError:
    waves = {}
# batches a PMC pass rendered: --rounds 1 --warmup 1 --steps 1 = two (rounds 2-4); since round 5 bench.py renders one more while it settles the samples in
# flight that fit the device (three).  Counted, not assumed: the any-hit kernel is launched four times per batch.
_anyhit = insts.get("k_trace<true>", {}).get("SQ_INSTS_VALU", (0, 8))[1]
STEPS = max(1, _anyhit // 4)
# bytes per unit that FETCH_SIZE misses: the 16-B-per-lane reads of consecutive queue entries, counted at 1/2
HALF_COUNTED = {
    "k_trace<true>": (16.0, "ray origin+length and direction+pixel, 2 x 16 B per ray at hand-out"),
    "k_trace<false>": (16.0, "ray origin and direction, 2 x 16 B per ray at hand-out"),
    "k_trace_packet<false>": (0.0, "generates its rays itself: no queue reads; node and triangle data through the scalar cache and L2"),
    "k_trace_multi<4>": (0.0, "generates its rays itself: no queue reads; node and triangle data through the scalar cache and L2"),
    "k_shade<false>": (24.0, "ray origin, direction and hit record, 3 x 16 B per entry (a bounce ray's throughput adds 8 B)"),
    "k_gen": (0.0, "writes only"),
}
out = {"scene_flags": bench["config"].get("scene_flags", 0),
       # the build the counters were taken on: bench.py compares it with the tree it runs on (roofline.traffic_stale)
       "csrc_sha256": bench["config"].get("csrc_sha256"),
       "config": {"width": bench["config"]["width"], "height": bench["config"]["height"], "level": bench["config"]["level"],
                  "samples_in_flight": bench["config"]["samples_in_flight"], "n_gpus": bench["n_gpus"]},
       "calibration": "FETCH_SIZE / WRITE_SIZE in KB, summed over the dispatches of one bench step; FETCH_SIZE counts 16 B/lane coalesced "
                      "reads at 1/2 (MI355X_MICROARCH.md; k_fold_planes -- (samples in flight - 1) planes x owned pixels x 16 B per launch -- reads back "
                      "at x0.50 in the same passes), everything else 1:1; WRITE_SIZE exact (primary rays x 32 B).",
       "kernels": {}, "issue": {},
       "issue_model": ("cycles a wave64 vector instruction occupies its SIMD's issue port, measured (tools/micro/valu_issue.hip, "
                       f"profiles/round2/r2c_valu_issue.md): full rate 2 (fma/mul/add, logic, shifts, moves), half rate 4 (min/max, compares, selects, "
                       "conversions, VOP3 integer, packed f32), quarter rate 8 (rcp/rsq/sqrt/exp/log); valu_issue_frac = SQ_INSTS_VALU x the mean "
                       f"cost of the kernel's inner-loop mix (profiles/{rnd}/isa_mix.json) / (1024 SIMDs x kernel cycles).  Round 5 "
                       "(profiles/round5/r5r_valu_issue_pairs.md): a plain FP32 fma / mul / add NEXT TO a half-rate instruction costs about half its "
                       "price (the pair 2.55 units against 2.0 + 1.12), so a mix of both can show valu_issue_frac above 1 under this additive model"),
       "source": [f"profiles/{rnd}/{tag}_{p}" for p in ("pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt", "pmc_SQ_INSTS_VALU.txt",
                                                          "pmc_GRBM_GUI_ACTIVE.txt", "pmc_SQ_WAVES.txt", "bench.json")]}
for name, k in kern.items():
    if name not in fetch and name not in write:
        continue
    units = k["units_per_launch"] * k["launches"] / bench["config"].get("batches_per_step", 1)  # per batch: a PMC pass runs --rounds 1
    rd = fetch.get(name, {}).get("FETCH_SIZE", (0, 0))[0] * 1024 / STEPS / units
    wr = write.get(name, {}).get("WRITE_SIZE", (0, 0))[0] * 1024 / STEPS / units
    add, why = HALF_COUNTED.get(name, (0.0, ""))
    out["kernels"][name] = {"units_per_batch": int(units), "raw_KB_per_batch": {"FETCH_SIZE": round(fetch.get(name, {}).get("FETCH_SIZE", (0, 0))[0] / STEPS),
                                                                          "WRITE_SIZE": round(write.get(name, {}).get("WRITE_SIZE", (0, 0))[0] / STEPS)},
                            "correction": f"+{add:g} B per unit read ({why})" if add else why,
                            "bytes_per_unit": {"read": round(rd + add, 1), "write": round(wr, 1), "total": round(rd + add + wr, 1)},
                            "algorithmic_bytes_per_unit": k["algorithmic_bytes_per_unit"]}
    if name in insts and name in act:
        cycles = act[name]["GRBM_GUI_ACTIVE"][0] / 8.0  # summed over the 8 XCDs
        simd_cycles = 1024.0 * cycles
        i = insts[name]
        cost = mix.get(name, {}).get("inner_loops", {}).get("avg_cycles", 4.0)
        e = {"valu_issue_frac": round(i["SQ_INSTS_VALU"][0] * cost / simd_cycles, 3), "valu_cycles_per_instruction": cost,
             "valu_instructions_per_unit": round(i["SQ_INSTS_VALU"][0] / STEPS / units, 2)}
        if "SQ_THREAD_CYCLES_VALU" in act[name] and "SQ_ACTIVE_INST_VALU" in act[name]:
            e["valu_active_lanes"] = round(act[name]["SQ_THREAD_CYCLES_VALU"][0] / (64.0 * act[name]["SQ_ACTIVE_INST_VALU"][0]), 3)
            # rocprof's own VALUBusy: SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles (a COUNTER, next to the modelled valu_issue_frac)
            e["valu_busy"] = round(act[name]["SQ_ACTIVE_INST_VALU"][0] * 4.0 / simd_cycles, 3)
        w = waves.get(name, {})
        if "SQ_WAVE_CYCLES" in w and w["SQ_WAVE_CYCLES"][0] > 0:  # what the resident waves were doing, as fractions of wave-cycles
            e["wave_cycles_issuing"] = round(w.get("SQ_ACTIVE_INST_ANY", (0, 0))[0] / w["SQ_WAVE_CYCLES"][0], 3)
            e["wave_cycles_waiting"] = round(w.get("SQ_WAIT_INST_ANY", (0, 0))[0] / w["SQ_WAVE_CYCLES"][0], 3)
        out["issue"][name] = e
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
