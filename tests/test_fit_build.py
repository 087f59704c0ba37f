"""checkFit (core.go:741-759) as a batched fit-mask build.

CPU: the C oracle (ids, CSR) against the independent string-level restatement (oracle/naive_fit.py)
through the interning marshaller, on hand-written known answers for each upstream rule and on seeded
random scenes.  GPU: bs_fit_build against the C oracle, bit-exact, and the masks it leaves in the
context drive the same decisions as masks loaded with bs_fit_load.
"""
import ctypes as C
import importlib

import numpy as np
import pytest

import naive_fit
import orc

pkg = importlib.import_module("batch-scheduler_amd")
soa = importlib.import_module("batch-scheduler_amd.soa")
fitspec = importlib.import_module("batch-scheduler_amd.fitspec")
synth = importlib.import_module("batch-scheduler_amd.synth")

NODE = {"name": "n1", "labels": {"zone": "a", "rack": "12", "disk": "ssd", "odd": "12a", "empty": ""},
        "taints": [("dedicated", "ml", "NoSchedule"), ("soft", "x", "PreferNoSchedule")]}
TOL = [("dedicated", "Equal", "ml", "NoSchedule")]


def tpl(sel=None, required=None, tolerations=TOL):
    return {"node_selector": sel or {}, "required": required, "tolerations": list(tolerations)}


def term(*exprs, fields=()):
    return {"expressions": list(exprs), "fields": list(fields)}


# (template, expected) on NODE — one line per upstream rule (U6.x in oracle/bs_oracle_fit.c)
KATS = [
    (tpl(), True),
    (tpl(sel={"zone": "a"}), True),
    (tpl(sel={"zone": "b"}), False),
    (tpl(sel={"zone": "a", "disk": "hdd"}), False),
    (tpl(sel={"missing": "a"}), False),
    (tpl(sel={"empty": ""}), True),
    (tpl(sel={"bad key!": "zzz", "zone": "b"}), True),            # U6.1: invalid pair -> empty selector matches all
    (tpl(sel={"zone": "not valid!"}), True),
    (tpl(required=[]), False),                                      # U6.2: no term matches
    (tpl(required=[term()]), False),                                # empty term selects nothing
    (tpl(required=[term(), term(("zone", "In", ["a"]))]), True),    # terms are ORed
    (tpl(required=[term(("zone", "In", ["b", "c"]))]), False),
    (tpl(required=[term(("zone", "In", ["b"]), ("disk", "Exists", []))]), False),   # ANDed inside a term
    (tpl(required=[term(("zone", "NotIn", ["b"]))]), True),
    (tpl(required=[term(("missing", "NotIn", ["b"]))]), True),      # U6.4: NotIn passes without the key
    (tpl(required=[term(("missing", "In", ["b"]))]), False),
    (tpl(required=[term(("disk", "Exists", []))]), True),
    (tpl(required=[term(("disk", "DoesNotExist", []))]), False),
    (tpl(required=[term(("missing", "DoesNotExist", []))]), True),
    (tpl(required=[term(("rack", "Gt", ["11"]))]), True),
    (tpl(required=[term(("rack", "Gt", ["12"]))]), False),
    (tpl(required=[term(("rack", "Lt", ["13"]))]), True),
    (tpl(required=[term(("rack", "Lt", ["+13"]))]), False),         # parses, but "+13" is not a valid label value
    (tpl(required=[term(("rack", "Gt", ["-1"]))]), False),          # same for a leading "-"
    (tpl(required=[term(("rack", "Gt", ["007"]))]), True),          # leading zeros parse as decimal
    (tpl(required=[term(("odd", "Gt", ["1"]))]), False),            # label value does not parse
    (tpl(required=[term(("missing", "Lt", ["1"]))]), False),
    (tpl(required=[term(("rack", "Gt", ["abc"]))]), False),         # U6.3 conversion errors skip the term
    (tpl(required=[term(("rack", "Gt", ["1", "2"]))]), False),
    (tpl(required=[term(("zone", "In", []))]), False),
    (tpl(required=[term(("disk", "Exists", ["ssd"]))]), False),
    (tpl(required=[term(("zone", "Foo", ["a"]))]), False),
    (tpl(required=[term(("bad key!", "DoesNotExist", []))]), False),
    (tpl(required=[term(("zone", "NotIn", ["not valid!"]))]), False),
    (tpl(required=[term(("zone", "In", ["a"]), ("rack", "Gt", ["abc"]))]), False),  # one bad requirement poisons the term
    (tpl(required=[term(("rack", "Gt", ["abc"])), term(("zone", "In", ["a"]))]), True),
    (tpl(required=[term(("empty", "In", [""]))]), True),
    (tpl(required=[term(fields=[("metadata.name", "In", ["n1"])])]), True),          # U6.5
    (tpl(required=[term(fields=[("metadata.name", "In", ["n2"])])]), False),
    (tpl(required=[term(fields=[("metadata.name", "NotIn", ["n2"])])]), True),
    (tpl(required=[term(fields=[("metadata.name", "In", ["n1", "n2"])])]), False),   # exactly one value
    (tpl(required=[term(fields=[("metadata.name", "Exists", [])])]), False),
    (tpl(required=[term(fields=[("other.field", "In", [""])])]), True),              # missing field reads ""
    (tpl(required=[term(fields=[("other.field", "NotIn", [""])])]), False),
    (tpl(required=[term(("zone", "In", ["a"]), fields=[("metadata.name", "In", ["n2"])])]), False),
    (tpl(sel={"zone": "a"}, required=[term(("zone", "In", ["b"]))]), False),
    (tpl(tolerations=[]), False),                                    # U6.6: NoSchedule taint not tolerated
    (tpl(tolerations=[("dedicated", "Equal", "web", "NoSchedule")]), False),
    (tpl(tolerations=[("dedicated", "", "ml", "")]), True),          # U6.7: "" operator is Equal, "" effect matches all
    (tpl(tolerations=[("dedicated", "Exists", "", "")]), True),
    (tpl(tolerations=[("", "Exists", "", "")]), True),               # empty key + Exists tolerates everything
    (tpl(tolerations=[("", "Equal", "ml", "")]), True),              # empty key skips the key comparison
    (tpl(tolerations=[("dedicated", "Exists", "", "NoExecute")]), False),
    (tpl(tolerations=[("dedicated", "Weird", "ml", "")]), False),
    (tpl(tolerations=[("dedicated", "Equal", "ml", "PreferNoSchedule")]), False),
    (tpl(tolerations=[("soft", "Equal", "x", ""), ("dedicated", "Equal", "ml", "NoSchedule")]), True),
]


@pytest.mark.parametrize("idx", range(len(KATS)))
def test_known_answers_naive_and_oracle(idx):
    t, exp = KATS[idx]
    assert naive_fit.check_fit(t, NODE) is exp
    nl, ft = fitspec.marshal([NODE], [t])
    assert orc.check_fit(nl, np.zeros(1, np.uint8), ft, 0, 0) is exp


def test_node_flags_force_no_fit():
    nl, ft = fitspec.marshal([NODE], [tpl()])
    for fl, exp in ((0, True), (soa.NODE_UNSCHEDULABLE, True), (soa.NODE_NIL, False), (soa.NODE_NO_NODE, False), (soa.NODE_TAINT_ERR, False)):
        assert orc.check_fit(nl, np.array([fl], np.uint8), ft, 0, 0) is exp
        assert naive_fit.check_fit(tpl(), NODE, fl) is exp


def test_go_parse_int_and_validation_agree():
    for s in ["0", "-0", "+5", "007", "12a", "", "+", "-", " 1", "1_000", "9223372036854775807", "9223372036854775808",
              "-9223372036854775808", "-9223372036854775809", "0x10", "1e3", "5\n", "\n5", "5\n\n"]:
        v, ok = fitspec.parse_int(s)
        assert (naive_fit.go_parse_int(s) is not None) == ok
        if ok:
            assert naive_fit.go_parse_int(s) == v
    for k in ["a", "a/b", "a/b/c", "", "/x", "x/", "Example.com/x", "example.com/x", "ex_ample.com/x", "a" * 63, "a" * 64,
              "-a", "a-", "a.b_c-d", "bad key!", "a\n", "example.com\n/x", "example.com/x\n", "x" * 253 + "/y", ("x" * 63 + ".") * 3 + "x" * 61 + "/y", ("x" * 63 + ".") * 4 + "/y"]:
        assert fitspec.label_key_ok(k) == naive_fit.qualified_name_ok(k), k
    for v in ["", "a", "a b", "a" * 63, "a" * 64, "_a", "a_", "A.b-C_d", "é", "a\n", "\n"]:
        assert fitspec.label_value_ok(v) == naive_fit.label_value_ok(v), v


def _flags(seed, n):
    rng = np.random.default_rng(seed)
    f = np.zeros(n, np.uint8)
    r = rng.random(n)
    f[r < 0.03] = soa.NODE_TAINT_ERR
    f[(r >= 0.03) & (r < 0.05)] = soa.NODE_NO_NODE
    f[(r >= 0.05) & (r < 0.06)] = soa.NODE_NIL
    f[(r >= 0.06) & (r < 0.10)] = soa.NODE_UNSCHEDULABLE
    return f


@pytest.mark.parametrize("seed,n,c", [(1, 1, 1), (2, 33, 7), (3, 64, 16), (4, 97, 40), (5, 200, 60), (6, 65, 25), (7, 300, 30), (8, 128, 50)])
def test_oracle_matches_naive_on_random_scenes(seed, n, c):
    nodes, templates = synth.make_fit_scene(seed, n, c)
    flags = _flags(seed, n)
    nl, ft = fitspec.marshal(nodes, templates)
    got = soa.FitMasks(orc.fit_build(nl, flags, ft), n).to_bool()
    exp = np.array([[naive_fit.check_fit(t, nd, int(flags[i])) for i, nd in enumerate(nodes)] for t in templates], dtype=bool)
    assert np.array_equal(got, exp)
    assert 0 < exp.mean() < 1 or n * c < 4, "scene should mix fits and misfits"


def test_scene_covers_every_rule():
    """The random scenes must exercise conversion errors, field selectors, every operator and every
    toleration shape, or the parity runs above prove little."""
    nodes, templates = synth.make_fit_scene(11, 50, 400)
    nl, ft = fitspec.marshal(nodes, templates)
    ops = set(int(x) for x in ft.exprs.op)
    assert {0, 1, 2, 3, 4, 5} <= {o & 0x7F for o in ops if not o & 0x80}
    assert any(o & 0x80 for o in ops)
    assert ft.fields.key.size and (ft.flags & fitspec.TPL_SELECTOR_INVALID).any() and (ft.flags & fitspec.TPL_HAS_REQUIRED).any()
    assert set(int(x) for x in ft.tol_op) >= {0, 1, 2, 3} and set(int(x) for x in ft.tol_effect) >= {0, 1, 2, 3, 4}
    assert (np.diff(ft.term_expr_off) == 0).any() and (np.diff(ft.term_off) == 0).any()


# ------------------------------------------------------------------------------------------- GPU
def _ctx_with_nodes(n, seed, scalars=1):
    capi = importlib.import_module("batch-scheduler_amd.capi")
    nodes = synth.make_nodes(seed, n, scalars, "warm")
    ctx = capi.Context(scalar_lanes=scalars)
    ctx.load_nodes(nodes)
    return ctx, nodes


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,c", [(1, 1, 1), (2, 31, 3), (3, 64, 5), (4, 65, 9), (5, 257, 33), (6, 1000, 64), (7, 5000, 200)])
def test_gpu_fit_build_matches_oracle(seed, n, c):
    ctx, nodes = _ctx_with_nodes(n, seed)
    scene_nodes, templates = synth.make_fit_scene(seed, n, c)
    nl, ft = fitspec.marshal(scene_nodes, templates)
    ctx.build_fit(nl, ft)
    got = ctx.read_fit()
    exp = orc.fit_build(nl, nodes.flags, ft)
    assert np.array_equal(got.bits, exp)


@pytest.mark.gpu
def test_gpu_fit_build_kats():
    ctx, nodes = _ctx_with_nodes(1, 3)
    flags = np.zeros(1, np.uint8)
    nodes.flags[:] = 0
    ctx.load_nodes(nodes)
    templates = [t for t, _ in KATS]
    nl, ft = fitspec.marshal([NODE], templates)
    ctx.build_fit(nl, ft)
    got = ctx.read_fit().to_bool()[:, 0]
    assert np.array_equal(got, np.array([e for _, e in KATS]))
    assert np.array_equal(ctx.read_fit().bits, orc.fit_build(nl, flags, ft))


@pytest.mark.gpu
def test_gpu_built_masks_drive_the_same_batch():
    """Masks built on the device == masks loaded through bs_fit_load: same decisions for a whole batch."""
    capi = importlib.import_module("batch-scheduler_amd.capi")
    nodes, fit, groups, pods, meta = synth.make("cfg2", "warm", seed=5)
    n, c, S = nodes.n, fit.n_classes, meta["scalars"]
    scene_nodes, templates = synth.make_fit_scene(9, n, c, quirks=False)
    nl, ft = fitspec.marshal(scene_nodes, templates)
    masks = soa.FitMasks(orc.fit_build(nl, nodes.flags, ft), n)
    outs = []
    for how in ("load", "build"):
        ctx = capi.Context(scalar_lanes=S)
        ctx.load_nodes(nodes)
        if how == "load":
            ctx.load_fit(masks)
        else:
            ctx.build_fit(nl, ft)
        ctx.load_groups(groups)
        ctx.load_pods(pods)
        outs.append(ctx.batch())
    for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "fl_bitmap", "group_admit", "group_ready"):
        assert np.array_equal(getattr(outs[0], name), getattr(outs[1], name)), name
    exp = orc.Sop(orc.Snapshot(nodes, masks, S), groups.copy()).batch(pods)
    assert np.array_equal(outs[1].pf_code, exp.pf_code) and np.array_equal(outs[1].fl_feasible, exp.fl_feasible)


@pytest.mark.gpu
def test_gpu_fit_build_errors():
    capi = importlib.import_module("batch-scheduler_amd.capi")
    ctx, nodes = _ctx_with_nodes(10, 1)
    nl, ft = fitspec.marshal(*synth.make_fit_scene(1, 9, 2))
    with pytest.raises(AssertionError):
        ctx.build_fit(nl, ft)
    st_n, st_t = nl.as_struct(), ft.as_struct()
    assert ctx._lib.bs_fit_build(ctx._h, C.byref(st_n), C.byref(st_t)) == -1      # BS_ERR_INVALID
    fresh = capi.Context(scalar_lanes=1)
    assert fresh._lib.bs_fit_build(fresh._h, C.byref(st_n), C.byref(st_t)) == -4  # BS_ERR_STATE


def test_oracle_fit_throughput_is_reported(capsys):
    """Times the C oracle's checkFit loop (the CPU figure quoted beside bs_fit_build in BASELINE.md)."""
    import time
    nodes, templates = synth.make_fit_scene(20260921, 2000, 100)
    nl, ft = fitspec.marshal(nodes, templates)
    flags = np.zeros(2000, np.uint8)
    t0 = time.perf_counter()
    bits = orc.fit_build(nl, flags, ft)
    dt = time.perf_counter() - t0
    assert bits.shape == (100, 63)
    with capsys.disabled():
        print(f"\n[oracle checkFit] {2000 * 100 / dt:.3e} pairs/s on one core")


# ---- property test: arbitrary (also malformed) strings through the marshaller and the C oracle must give what the
# string-level restatement gives.  Strategies are biased towards the characters the upstream validators care about.
from hypothesis import given, settings, strategies as st_

_key_chars = "abzAZ09-_./ !\n"
_keys = st_.one_of(st_.sampled_from(["zone", "rack", "disk", "a/b", "example.com/x", "bad key!", "", "x" * 64, "a/b/c", "-a", "k8s.io/role", "zone\n", "example.com\n/x"]),
                   st_.text(alphabet=_key_chars, min_size=0, max_size=6))
_vals = st_.one_of(st_.sampled_from(["", "a", "b", "1", "12", "007", "+5", "-3", "12a", "9223372036854775807", "9223372036854775808", "no good", "x" * 64, "5\n", "a\n"]),
                   st_.text(alphabet="ab019+-_. \n", min_size=0, max_size=5))
_ops = st_.sampled_from(["In", "NotIn", "Exists", "DoesNotExist", "Gt", "Lt", "Foo", ""])
_effects = st_.sampled_from(["NoSchedule", "NoExecute", "PreferNoSchedule", "", "Other"])
_tol_ops = st_.sampled_from(["", "Equal", "Exists", "Weird"])
_expr = st_.tuples(_keys, _ops, st_.lists(_vals, max_size=3))
_field = st_.tuples(st_.sampled_from(["metadata.name", "metadata.namespace", ""]), st_.sampled_from(["In", "NotIn", "Exists", "Bogus"]),
                    st_.lists(st_.sampled_from(["n0", "n1", "", "default"]), max_size=2))
_term = st_.fixed_dictionaries({"expressions": st_.lists(_expr, max_size=3), "fields": st_.lists(_field, max_size=2)})
_template = st_.fixed_dictionaries({
    "node_selector": st_.dictionaries(_keys, _vals, max_size=2),
    "required": st_.one_of(st_.none(), st_.lists(_term, max_size=3)),
    "tolerations": st_.lists(st_.tuples(_keys, _tol_ops, _vals, _effects), max_size=3)})
_node = st_.fixed_dictionaries({
    "name": st_.sampled_from(["n0", "n1", "n2", ""]),
    "labels": st_.dictionaries(_keys, _vals, max_size=4),
    "taints": st_.lists(st_.tuples(_keys, _vals, _effects), max_size=3)})


@settings(max_examples=300, deadline=None)
@given(nodes=st_.lists(_node, min_size=1, max_size=4), templates=st_.lists(_template, min_size=1, max_size=3),
       flags=st_.lists(st_.sampled_from([0, 0, 0, soa.NODE_UNSCHEDULABLE, soa.NODE_TAINT_ERR, soa.NODE_NO_NODE, soa.NODE_NIL]), min_size=4, max_size=4))
def test_fit_property_oracle_matches_string_level(nodes, templates, flags):
    fl = np.array(flags[: len(nodes)], np.uint8)
    nl, ft = fitspec.marshal(nodes, templates)
    got = soa.FitMasks(orc.fit_build(nl, fl, ft), len(nodes)).to_bool()
    exp = np.array([[naive_fit.check_fit(t, nd, int(fl[i])) for i, nd in enumerate(nodes)] for t in templates], dtype=bool)
    assert np.array_equal(got, exp)


@pytest.mark.gpu
@settings(max_examples=150, deadline=None)
@given(nodes=st_.lists(_node, min_size=1, max_size=4), templates=st_.lists(_template, min_size=1, max_size=3),
       flags=st_.lists(st_.sampled_from([0, 0, 0, soa.NODE_UNSCHEDULABLE, soa.NODE_TAINT_ERR, soa.NODE_NO_NODE, soa.NODE_NIL]), min_size=4, max_size=4))
def test_gpu_fit_property_matches_oracle(nodes, templates, flags):
    capi = importlib.import_module("batch-scheduler_amd.capi")
    n = len(nodes)
    base = synth.make_nodes(1, n, 0, "warm")
    base.flags[:] = np.array(flags[:n], np.uint8)
    nl, ft = fitspec.marshal(nodes, templates)
    ctx = _PROP_CTX.get("ctx")
    if ctx is None:
        ctx = _PROP_CTX["ctx"] = capi.Context(scalar_lanes=0)
    ctx.load_nodes(base)
    ctx.build_fit(nl, ft)
    assert np.array_equal(ctx.read_fit().bits, orc.fit_build(nl, base.flags, ft))


_PROP_CTX = {}
