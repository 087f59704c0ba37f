"""CPU: register / scratch footprint of the kernels on the hot paths, read from the built code object (tools/kernel_resources.py).

A kernel that touches scratch memory (`.private_segment_fixed_size` != 0) pays for the scratch set-up on every launch — DESIGN.md
section 4 has two measured cases (57 instead of 23 us, 75 instead of 57 us) —, and it gets there silently: an indexed access to a
register array, a pointer that may name two structs, one spilled value.  The hot kernels of the three chains, of the queue patch and
of the throughput regime must stay at zero; the known exceptions are listed with their reason."""
import importlib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HOT = ("k_fast_query_tables", "k_fast_scan_filter_final", "k_fast_scan_filter", "k_fast_scan", "k_fast_filter", "k_fast_final", "k_fast_commit",
       "k_epoch_query_tables", "k_epoch_scan_filter", "k_epoch_final", "k_pods_apply", "k_leader_info", "k_filter", "k_scan", "k_query", "k_tally", "k_ready",
       "k_fd_apply", "k_fd_events", "k_nodes_derive", "k_filter_expand")
# kernel (demangled prefix) -> bytes of scratch it is known to use, and why that is tolerated
KNOWN = {
    "k_seq_pass": "one persistent launch per pass (20 bytes; the generic-lane instantiation 1492): the set-up is paid once per pass (tens of ms)",
    "k_commit": "general chain's commit pass only (144 bytes)",
    "k_fast_scan_filter_t<1>": "the throughput regime's one-launch form at one scalar lane, compiled for FOUR waves per SIMD (126 VGPRs, nine dwords spilled = 36 bytes): "
                               "the launch lasts 20-300 us and is bound by what its waves keep in flight; measured faster with the fourth wave than without the spill "
                               "(bs_fast.hpp, profiles/r06_waves4_ab.txt)",
}


def _demangle(name: str) -> str:
    m = re.match(r"_ZN2bs(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[len(m.group(0)):len(m.group(0)) + n]
    t = re.match(r"ILi(n?)(\d+)E", name[len(m.group(0)) + n:])
    return base + (f"<{'-' if t.group(1) else ''}{t.group(2)}>" if t else "")


def test_hot_kernels_use_no_scratch(bsa):
    import kernel_resources as kr
    res = kr.resources(bsa.build.build())
    assert len(res) >= 200, len(res)
    offenders = {}
    seen_hot = set()
    for mangled, r in res.items():
        name = _demangle(mangled)
        base = name.split("<")[0]
        if base in HOT:
            seen_hot.add(base)
        if r["scratch"] and not any(name == k or base == k for k in KNOWN):
            offenders[name] = r["scratch"]
    assert not offenders, f"kernels that use scratch memory: {offenders}"
    assert seen_hot == set(HOT), set(HOT) - seen_hot          # (the list names kernels that exist)
    # the exceptions are still exceptions (when one goes away, take it off the list)
    for k in KNOWN:
        assert any(_demangle(m) == k or _demangle(m).split("<")[0] == k for m, r in res.items() if r["scratch"]), f"{k} no longer uses scratch: update KNOWN"


def test_throughput_regime_kernels_keep_their_footprint(bsa):
    """The transposed Filter item on its own (BS_TP_FILTER=5): no LDS, no scratch, five waves per SIMD since round 6 (two tiles of request slots per
    wave: two request sets, 2 k accumulators and two row pointers per lane — 100 VGPRs; 64-66 up to round 5 with one tile).  In one launch with the
    scan role (the default) the kernel stays at the scan's footprint, which is what decides its waves per SIMD."""
    import kernel_resources as kr
    res = {_demangle(m): r for m, r in kr.resources(bsa.build.build()).items()}
    ft = res["k_fast_filter_t"]
    assert ft["vgpr"] <= 102 and ft["lds"] == 0 and ft["scratch"] == 0, ft            # 102 = the five-wave line (512 / 5)
    for s in range(0, 5):
        both, scan = res[f"k_fast_scan_filter_t<{s}>"], res[f"k_fast_scan<{s}>"]
        waves = lambda v: 512 // (-(-v // 8) * 8)                                      # waves per SIMD the VGPR count admits (granule 8)
        assert waves(both["vgpr"]) >= waves(scan["vgpr"]) and both["vgpr"] <= scan["vgpr"] + 4 and (both["scratch"] == 0 or s == 1), (s, both, scan)
