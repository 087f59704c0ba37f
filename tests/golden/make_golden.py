"""Generates tests/golden/*.json.  Run in the build container (x86-64); commit the outputs.

1. core_test_vectors.json — the reference's ONLY result-pinning test for the hot path,
   /root/reference/pkg/scheduler/core/core_test.go:27-115, transcribed to int64 lanes
   (node :50-68, resident pod :28-48 / AddPod :72, cases :82-103, assertion :108-112).
2. f32_scale_kat.json — known answers for int64(float32(a)*pct) (core.go:656-659,667) computed
   with numpy float32 on x86-64 SSE2 — the same IEEE-754 operations Go/amd64 emits (CVTSQ2SS /
   MULSS / CVTTSS2SQ).  Inputs below 2**53 only, so numpy's int -> double -> float32 path is a
   single rounding; includes the values listed in SURVEY.md §8(c).
3. readme_race_scene.json — the README's 2 gangs x 5 pods on one 8-CPU node
   (README.md:78-188) as the per-decision walk-through of SURVEY.md §8(c).
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def scale_np(a: int, pct: float) -> int:
    assert abs(a) < 2 ** 53
    return int(np.float32(np.float32(a) * np.float32(pct)))


def main():
    core_test = {
        "source": "pkg/scheduler/core/core_test.go:27-115",
        "scalar_names": ["alpha.kubernetes.io/nvidia-gpu", "tencent.cr/tencentip"],
        "node": {"allocatable": {"cpu": 10000, "pods": 100, "alpha.kubernetes.io/nvidia-gpu": 10, "tencent.cr/tencentip": 20},
                 "requested": {"cpu": 1000, "alpha.kubernetes.io/nvidia-gpu": 1, "tencent.cr/tencentip": 1},
                 "pod_count": 1},
        "percent": 1.0,
        "expected_left": {"cpu": 9000, "memory": 0, "ephemeral-storage": 0, "pods": 99,
                          "alpha.kubernetes.io/nvidia-gpu": 9, "tencent.cr/tencentip": 19},
        "cases": [
            {"req": {"cpu": 1000, "alpha.kubernetes.io/nvidia-gpu": 1, "tencent.cr/tencentip": 1}, "desire": True},
            {"req": {"cpu": 1000, "alpha.kubernetes.io/nvidia-gpu": 101, "tencent.cr/tencentip": 1}, "desire": False},
            {"req": {"cpu": 1000, "alpha.kubernetes.io/nvidia-gpu": 1, "tencent.cr/tencentip": 101}, "desire": False},
        ],
    }
    json.dump(core_test, open(os.path.join(HERE, "core_test_vectors.json"), "w"), indent=1)

    listed = [(8000, .7), (110, .7), (100, .7), (16777217, 1), (16777219, 1), (16777219, .7), (16655429632, 1),
              (16655429632, .7), (270255247360, 1), (270255247360, .7), (2 ** 34 + 1024, 1), (2 ** 34 + 1025, 1),
              (540510494720, 1), (1099511623679, 1), (7, .7), (3, .7), (1, .7), (0, .7), (0, 1)]
    rng = np.random.default_rng(20260921)
    extra = []
    for bits in range(1, 53):
        for _ in range(24):
            a = int(rng.integers(1 << (bits - 1), 1 << bits))
            extra.append((a, 1.0))
            extra.append((a, 0.7))
            extra.append((-a, 0.7))
    for e in range(24, 52):           # ties and neighbours at every float32 ulp size
        ulp = 1 << (e - 23)
        base = (1 << e) + 5 * ulp
        for d in (-1, 0, 1):
            for half in (ulp // 2, ulp + ulp // 2):
                extra.append((base + half + d, 1.0))
                extra.append((base + half + d, 0.7))
    for pct in (0.5, 0.3, 0.9, 1.5):
        for a in (1000, 123456789, 2 ** 40 + 12345):
            extra.append((a, pct))
    kat = [{"a": a, "pct_bits": int(np.float32(p).view(np.uint32)), "out": scale_np(a, p)} for a, p in listed + extra]
    json.dump({"source": "numpy float32 on x86-64 (see docstring)", "listed": len(listed), "vectors": kat},
              open(os.path.join(HERE, "f32_scale_kat.json"), "w"))

    race = {
        "source": "README.md:78-188; SURVEY.md 8(c) walk-through of core.go:88-167",
        "node": {"allocatable_cpu": 8000, "allocatable_pods": 110, "requested_cpu": 900, "pod_count": 9},
        "groups": {"group1": {"min_member": 5}, "group2": {"min_member": 5}},
        "pod_cpu": 1000,
        "steps": [
            {"what": "first group1 pod, nothing matched: branch B, need 5x1000 <= scale(8000,1)-900 = 7100",
             "pod_group": "group1", "state": {"group1": {"matched": 0, "has_pod": False}, "group2": {"matched": 0, "has_pod": False}},
             "node_requested_cpu": 900, "expect_code": "PASS_FIRST_FITS", "expect_first_k": 0},
            {"what": "later group1 pod, group1 is the leader: branch C",
             "pod_group": "group1", "state": {"group1": {"matched": 2, "has_pod": True}, "group2": {"matched": 0, "has_pod": False}},
             "node_requested_cpu": 2900, "expect_code": "PASS_IS_MAX"},
            {"what": "group2 pod after two group1 pods were assumed: branch D, need 3x1000+1000 vs scale(8000,.7)-2900 = 2700",
             "pod_group": "group2", "state": {"group1": {"matched": 2, "has_pod": True}, "group2": {"matched": 0, "has_pod": False}},
             "node_requested_cpu": 2900, "expect_code": "REJECT_RESERVE"},
            {"what": "group1 latched Scheduled; group2 alone: branch B, need 5000 vs 8000-5900 = 2100",
             "pod_group": "group2", "state": {"group1": {"matched": 5, "has_pod": True, "scheduled_latch": True}, "group2": {"matched": 0, "has_pod": False}},
             "node_requested_cpu": 5900, "expect_code": "REJECT_FIRST"},
        ],
        "expected_end_state": {"group1": "5/5 admitted", "group2": "0/5"},
    }
    json.dump(race, open(os.path.join(HERE, "readme_race_scene.json"), "w"), indent=1)
    print("wrote", len(kat), "scale vectors")


if __name__ == "__main__":
    main()
