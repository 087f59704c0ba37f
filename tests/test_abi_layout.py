"""The ctypes structures of the Python binding against the C header, field by field: a C program compiled from
include/bsched.h prints sizeof / offsetof for every ABI struct, and every ctypes field has to sit at the same offset with the
same size.  (The Go side includes the header through cgo and cannot drift; the Python side restates it.)"""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bsa = importlib.import_module("batch-scheduler_amd")
capi, soa, fitspec = bsa.capi, bsa.soa, bsa.fitspec

PAIRS = [
    ("bs_config", capi.Config), ("bs_node_delta", capi.NodeDelta), ("bs_timing", capi.Timing), ("bs_batch_stats", capi.BatchStats),
    ("bs_nodes_soa", soa.NodesStruct), ("bs_groups_soa", soa.GroupsStruct), ("bs_pods_soa", soa.PodsStruct),
    ("bs_batch_out", soa.BatchOutStruct), ("bs_batch_view", soa.BatchViewStruct), ("bs_group_delta", soa.GroupDelta), ("bs_pods_delta", soa.PodsDeltaStruct), ("bs_pods_out", soa.PodsOutStruct),
    ("bs_node_request", capi.NodeRequest), ("bs_seq_out", capi.SeqOut),
    ("bs_node_labels", fitspec.NodeLabelsStruct), ("bs_requirements", fitspec.RequirementsStruct), ("bs_fit_templates", fitspec.FitTemplatesStruct),
]


def _c_fields(header: str, name: str):
    """field names of `typedef struct <name> { ... } <name>;` in declaration order (arrays and pointers included)"""
    body = re.search(r"typedef struct %s\s*\{(.*?)\}\s*%s\s*;" % (name, name), header, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
            fields.append(m.group(1))
    return fields


def test_ctypes_structures_match_the_header(tmp_path):
    header = open(os.path.join(ROOT, "include", "bsched.h")).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "bsched.h"', "int main(void) {"]
    for cname, _ in PAIRS:
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for f in _c_fields(header, cname):
            lines.append(f'  printf("{cname} {f} %zu %zu\\n", offsetof({cname}, {f}), sizeof((({cname}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    c_layout = {}
    for ln in out.splitlines():
        p = ln.split()
        if p[1] == "size":
            c_layout.setdefault(p[0], {})["__size__"] = int(p[2])
        else:
            c_layout.setdefault(p[0], {})[p[1]] = (int(p[2]), int(p[3]))
    for cname, ct in PAIRS:
        want = c_layout[cname]
        assert C.sizeof(ct) == want["__size__"], f"{cname}: sizeof {C.sizeof(ct)} != {want['__size__']}"
        py_fields = [f[0] for f in ct._fields_]
        c_fields = [k for k in want if k != "__size__"]
        assert py_fields == c_fields, f"{cname}: field order / names differ\n  python {py_fields}\n  header {c_fields}"
        for fname in py_fields:
            d = getattr(ct, fname)
            assert (d.offset, d.size) == want[fname], f"{cname}.{fname}: ctypes (offset, size) {(d.offset, d.size)} != header {want[fname]}"


def test_every_environment_switch_of_the_library_is_documented():
    """Every `getenv("BS_...")` in the library's sources appears in INTEGRATION.md's table of diagnostic switches (none of them is needed
    in production; a switch nobody can find is a behaviour nobody can explain)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for pat in ("batch-scheduler_amd/csrc/*", "batch-scheduler_amd/host/*"):
        for f in glob.glob(os.path.join(root, pat)):
            names |= set(re.findall(r'getenv\("(BS_[A-Z0-9_]+)"\)', open(f).read()))
    assert len(names) >= 20, names
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, f"undocumented switches: {missing}"
