"""The flat-argument forms of the struct-taking entry points (what the cgo binding calls: every array its own argument, no
Go-allocated struct of Go pointers crosses by pointer) against the struct forms: same loads, same reads, same batch, same
sequential pass, same fit masks.  go/c11_client/shim_client.c runs the shim's cycle through them (tests/test_c11_client.py);
here every one of them is called once with ctypes."""
import ctypes as C
import importlib

import numpy as np
import pytest

from test_gpu_parity import load_ctx

pytestmark = pytest.mark.gpu
fitspec = importlib.import_module("batch-scheduler_amd.fitspec")
synth = importlib.import_module("batch-scheduler_amd.synth")


def fields(st):
    return [getattr(st, f[0]) for f in st._fields_]


def test_flat_forms_equal_struct_forms(bsa, soa, orc):
    capi = bsa.capi
    lib = capi.load_library()
    nodes, fit, groups, pods, _ = bsa.synth.make("cfg2", "cold")
    pods = pods.take(np.argsort(pods.group, kind="stable"))
    L = nodes.lanes
    with load_ctx(bsa, nodes, fit, groups, pods) as ref, bsa.Context(scalar_lanes=L - 4) as ctx:
        chk = ctx._chk
        ns, gs, ps = nodes.as_struct(), groups.as_struct(), pods.as_struct()
        chk(lib.bs_nodes_load_flat(ctx._h, *fields(ns)), "bs_nodes_load_flat")
        ctx.n = nodes.n
        ctx.load_fit(fit)
        chk(lib.bs_groups_load_flat(ctx._h, *fields(gs)), "bs_groups_load_flat")
        ctx.g = groups.g
        chk(lib.bs_pods_load_flat(ctx._h, *fields(ps)), "bs_pods_load_flat")
        ctx.p = pods.p
        # reads
        back = soa.Groups.empty(groups.g, L)
        chk(lib.bs_groups_read_flat(ctx._h, *fields(back.as_struct())), "bs_groups_read_flat")
        for name in ("min_member", "status_scheduled", "matched", "flags", "min_resources", "occupied_by"):
            assert np.array_equal(getattr(back, name), getattr(groups, name)), name
        pb = soa.Pods.empty(pods.p, L)
        s2 = pb.as_struct()
        chk(lib.bs_pods_read_flat(ctx._h, pods.p, s2.group, s2.req, s2.req_present, s2.cls, s2.owner, s2.flags), "bs_pods_read_flat")
        assert pb.equal(pods)
        # batch + read
        ctx.run(soa.STAGE_ALL)
        exp = ref.batch(soa.STAGE_ALL, bitmap=False, rows=True)
        out = soa.BatchOut.alloc(pods.p, groups.g, nodes.n, bitmap=False, rows_cap=max(ctx.filter_rows_count(), 1))
        chk(lib.bs_batch_read_flat(ctx._h, *fields(out.as_struct())), "bs_batch_read_flat")
        for name in ("pf_code", "pf_first_k", "pf_leader", "fl_code", "fl_feasible", "group_admit", "group_ready"):
            assert np.array_equal(getattr(out, name), getattr(exp, name)), name
        # queue patch
        rem = np.arange(3, dtype=np.uint32)
        ins = pods.take(np.arange(3, 6))
        i2 = ins.as_struct()
        chk(lib.bs_pods_apply_flat(ctx._h, 3, rem.ctypes.data_as(C.POINTER(C.c_uint32)), 0, None, None, ins.p, i2.group, i2.req, i2.req_present, i2.cls, i2.owner,
                                   i2.flags, None), "bs_pods_apply_flat")
        ref.apply_pods(remove=rem, insert=ins)
        assert ctx.read_pods().equal(ref.read_pods())
        # the sequential pass
        p = ctx.pods_count()
        cap = groups.g
        pf, node = np.zeros(p, np.uint8), np.zeros(p, np.int32)
        rg, rp = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        t0, t1, sc = np.zeros(cap, np.int64), np.zeros(cap, np.int64), np.zeros(8, np.int64)
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        u32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        chk(lib.bs_seq_run_flat(ctx._h, soa.STAGE_PREFILTER, pf.ctypes.data_as(C.POINTER(C.c_uint8)), None, None, node.ctypes.data_as(C.POINTER(C.c_int32)), cap,
                                u32(rg), u32(rp), i64(t0), i64(t1), i64(sc), None), "bs_seq_run_flat")
        r = ref.seq_run(soa.STAGE_PREFILTER)
        k = r["n_released"]
        assert sc[0] == k > 0 and np.array_equal(pf, r["pf_code"]) and np.array_equal(node, r["pod_node"]) and np.array_equal(rg[:k], r["released_group"])
        assert np.array_equal(ctx.read_node_requests()[0], ref.read_node_requests()[0])


def test_fit_build_flat_equals_struct_form(bsa, soa):
    lib = bsa.capi.load_library()
    nodes, _, _, _, _ = synth.make("cfg2", "cold")
    scene_nodes, templates = synth.make_fit_scene(11, nodes.n, 9)
    nl, ft = fitspec.marshal(scene_nodes, templates)
    with bsa.Context(scalar_lanes=nodes.lanes - 4) as a, bsa.Context(scalar_lanes=nodes.lanes - 4) as b:
        a.load_nodes(nodes)
        b.load_nodes(nodes)
        a.build_fit(nl, ft)
        ns, ts = nl.as_struct(), ft.as_struct()
        args = fields(ns) + [ts.c, ts.field_name_key, ts.flags, ts.sel_off, ts.sel_key, ts.sel_val, ts.term_off, ts.term_expr_off, ts.term_field_off]
        args += fields(ts.exprs) + fields(ts.fields) + [ts.tol_off, ts.tol_key, ts.tol_val, ts.tol_op, ts.tol_effect]
        b._chk(lib.bs_fit_build_flat(b._h, *args), "bs_fit_build_flat")
        b.n_classes = ft.c
        assert np.array_equal(a.read_fit().bits, b.read_fit().bits)
