"""bench.py quotes hardware counters (HBM bytes per unit, issue-port figures) from committed PMC passes (profiles/roundN/traffic*.json).  They are tied to
the build they were taken on: a fingerprint of csrc/ + the device compiler flags travels from bench.py's line (config.csrc_sha256) through
tools/traffic_json.py into the json, and bench.py holds it against the tree it runs on (a stale file was quoted once)."""
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "opencl-path-tracer_amd"))

import bench  # noqa: E402
from ptamd import build as B  # noqa: E402


def test_a_doctored_byte_of_csrc_flips_traffic_stale(tmp_path):
    copy = tmp_path / "csrc"
    shutil.copytree(B.CSRC_DIR, copy, ignore=shutil.ignore_patterns("*.so", "variants"))
    fp = B.csrc_fingerprint(str(copy))
    assert fp == B.csrc_fingerprint() == bench.csrc_sha256() and len(fp) == 64
    tj = {"csrc_sha256": fp, "kernels": {"k_trace<true>": {"bytes_per_unit": {"total": 100.0}}}, "issue": {"k_trace<true>": {"valu_busy": 1.1, "valu_active_lanes": 0.7}}}
    assert not bench.traffic_is_stale(tj, fp)
    k = bench.apply_traffic({"units_per_launch": 1000}, "k_trace<true>", tj, False)
    assert k["traffic"] == 100000 and k["valu_busy"] == 1.1 and k["valu_active_lanes"] == 0.7
    # one byte of one kernel source changes: the fingerprint moves, the counters are flagged and the issue-port figures dropped
    src = copy / "pt_trace.h"
    data = bytearray(src.read_bytes())
    data[len(data) // 2] ^= 1
    src.write_bytes(bytes(data))
    fp2 = B.csrc_fingerprint(str(copy))
    assert fp2 != fp
    assert bench.traffic_is_stale(tj, fp2)
    k = bench.apply_traffic({"units_per_launch": 1000}, "k_trace<true>", tj, True)
    assert k["traffic"] == 100000 and "valu_busy" not in k and "valu_active_lanes" not in k
    # a json that does not say which build it was taken on (rounds 1-5) is stale by definition; no json at all is not "stale", it is absent
    assert bench.traffic_is_stale({"kernels": {}}, fp) and not bench.traffic_is_stale(None, fp)
    # a new file in csrc/ counts too
    (copy / "pt_new.h").write_text("// nothing\n")
    assert B.csrc_fingerprint(str(copy)) not in (fp, fp2)


def test_committed_traffic_files_name_their_build_or_are_flagged():
    """Every traffic*.json bench.py may pick up either carries the fingerprint of a build (round 6 on) or is treated as stale."""
    import glob
    for path in glob.glob(os.path.join(ROOT, "profiles", "round*", "traffic*.json")):
        tj = json.load(open(path))
        rnd = int(os.path.basename(os.path.dirname(path)).replace("round", ""))
        if rnd >= 6:
            assert isinstance(tj.get("csrc_sha256"), str) and len(tj["csrc_sha256"]) == 64, path
        else:
            assert bench.traffic_is_stale(tj, bench.csrc_sha256()), path
