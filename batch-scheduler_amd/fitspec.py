"""Host-side marshalling for the fit-mask builder (``bs_fit_build``, include/bsched.h).

checkFit (core.go:741-759) matches a pod template's node selector / required node affinity /
tolerations against a node's labels and taints.  All of that is string work upstream; at the C ABI
only interned ids cross.  This module is what the Go shim would do on its side of the boundary:

* ``Interner``     string -> id, 0 reserved for "" (equal strings <=> equal ids);
* ``parse_int``    strconv.ParseInt(s, 10, 64) -> (value, ok);
* ``label_key_ok`` / ``label_value_ok``  apimachinery's IsQualifiedName / IsValidLabelValue, which
  decide BS_OP_INVALID and BS_TPL_SELECTOR_INVALID;
* ``NodeLabels`` / ``FitTemplates``  the CSR containers with ``as_struct()``;
* ``marshal(nodes, templates)``  object form -> containers.

Object form (what tests and the synthetic generator build):

    node     = {"name": str, "labels": {key: value}, "taints": [(key, value, effect), ...]}
    template = {"node_selector": {key: value},
                "required": None | [{"expressions": [(key, op, [values])], "fields": [(key, op, [values])]}, ...],
                "tolerations": [(key, operator, value, effect), ...]}
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass

import numpy as np

EFFECT_NONE, EFFECT_NO_SCHEDULE, EFFECT_PREFER_NO_SCHEDULE, EFFECT_NO_EXECUTE = 0, 1, 2, 3
TOL_OP_DEFAULT, TOL_OP_EQUAL, TOL_OP_EXISTS = 0, 1, 2
OP_IN, OP_NOT_IN, OP_EXISTS, OP_DOES_NOT_EXIST, OP_GT, OP_LT, OP_INVALID = 0, 1, 2, 3, 4, 5, 0x80
TPL_HAS_REQUIRED, TPL_SELECTOR_INVALID = 1, 2

_EFFECTS = {"": EFFECT_NONE, "NoSchedule": EFFECT_NO_SCHEDULE, "PreferNoSchedule": EFFECT_PREFER_NO_SCHEDULE,
            "NoExecute": EFFECT_NO_EXECUTE}
_TOL_OPS = {"": TOL_OP_DEFAULT, "Equal": TOL_OP_EQUAL, "Exists": TOL_OP_EXISTS}
_OPS = {"In": OP_IN, "NotIn": OP_NOT_IN, "Exists": OP_EXISTS, "DoesNotExist": OP_DOES_NOT_EXIST, "Gt": OP_GT, "Lt": OP_LT}

_INT_RE = re.compile(r"[+-]?[0-9]+")
_NAME_RE = re.compile(r"([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]")
_DNS_LABEL = r"[a-z0-9]([-a-z0-9]*[a-z0-9])?"
_SUBDOMAIN_RE = re.compile(_DNS_LABEL + r"(\." + _DNS_LABEL + r")*")


def parse_int(s: str):
    """strconv.ParseInt(s, 10, 64): optional sign, decimal digits only, must fit int64."""
    if not _INT_RE.fullmatch(s):
        return 0, False
    v = int(s)
    if v < -(1 << 63) or v > (1 << 63) - 1:
        return 0, False
    return v, True


def label_key_ok(key: str) -> bool:
    """validation.IsQualifiedName: [dns-subdomain '/'] name, name 1..63 chars of [A-Za-z0-9_.-] with
    alphanumeric ends, prefix a DNS-1123 subdomain of at most 253 chars."""
    parts = key.split("/")
    if len(parts) == 1:
        name = parts[0]
    elif len(parts) == 2:
        prefix, name = parts
        if not prefix or len(prefix) > 253 or not _SUBDOMAIN_RE.fullmatch(prefix):
            return False
    else:
        return False
    return 0 < len(name) <= 63 and bool(_NAME_RE.fullmatch(name))


def label_value_ok(value: str) -> bool:
    """validation.IsValidLabelValue: empty, or at most 63 chars shaped like a qualified-name part."""
    return value == "" or (len(value) <= 63 and bool(_NAME_RE.fullmatch(value)))


class Interner:
    """Equal strings <=> equal ids; id 0 is the empty string."""

    def __init__(self):
        self.ids = {"": 0}

    def __call__(self, s: str) -> int:
        i = self.ids.get(s)
        if i is None:
            i = self.ids[s] = len(self.ids)
        return i

    @staticmethod
    def code(s: str, table: dict, other: int) -> int:
        """Small enumerations (effects, toleration operators): known strings get their fixed code, any
        other string the catch-all `other` (such values never compare equal to a code that matters:
        only NoSchedule / NoExecute taints are looked at, unknown operators tolerate nothing)."""
        return table.get(s, other)


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class NodeLabelsStruct(C.Structure):
    _fields_ = [("n", C.c_uint32), ("name", C.POINTER(C.c_uint32)),
                ("label_off", C.POINTER(C.c_uint32)), ("label_key", C.POINTER(C.c_uint32)), ("label_val", C.POINTER(C.c_uint32)),
                ("label_int", C.POINTER(C.c_int64)), ("label_int_ok", C.POINTER(C.c_uint8)),
                ("taint_off", C.POINTER(C.c_uint32)), ("taint_key", C.POINTER(C.c_uint32)), ("taint_val", C.POINTER(C.c_uint32)),
                ("taint_effect", C.POINTER(C.c_uint8))]


class RequirementsStruct(C.Structure):
    _fields_ = [("count", C.c_uint32), ("key", C.POINTER(C.c_uint32)), ("op", C.POINTER(C.c_uint8)),
                ("val_off", C.POINTER(C.c_uint32)), ("val", C.POINTER(C.c_uint32)),
                ("val_int", C.POINTER(C.c_int64)), ("val_int_ok", C.POINTER(C.c_uint8))]


class FitTemplatesStruct(C.Structure):
    _fields_ = [("c", C.c_uint32), ("field_name_key", C.c_uint32), ("flags", C.POINTER(C.c_uint8)),
                ("sel_off", C.POINTER(C.c_uint32)), ("sel_key", C.POINTER(C.c_uint32)), ("sel_val", C.POINTER(C.c_uint32)),
                ("term_off", C.POINTER(C.c_uint32)), ("term_expr_off", C.POINTER(C.c_uint32)), ("term_field_off", C.POINTER(C.c_uint32)),
                ("exprs", RequirementsStruct), ("fields", RequirementsStruct),
                ("tol_off", C.POINTER(C.c_uint32)), ("tol_key", C.POINTER(C.c_uint32)), ("tol_val", C.POINTER(C.c_uint32)),
                ("tol_op", C.POINTER(C.c_uint8)), ("tol_effect", C.POINTER(C.c_uint8))]


def _u32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint32))


def _u8(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint8))


def _i64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int64))


@dataclass
class NodeLabels:
    name: np.ndarray
    label_off: np.ndarray
    label_key: np.ndarray
    label_val: np.ndarray
    label_int: np.ndarray
    label_int_ok: np.ndarray
    taint_off: np.ndarray
    taint_key: np.ndarray
    taint_val: np.ndarray
    taint_effect: np.ndarray

    @property
    def n(self) -> int:
        return len(self.name)

    def as_struct(self) -> NodeLabelsStruct:
        return NodeLabelsStruct(self.n, _p(self.name, C.c_uint32), _p(self.label_off, C.c_uint32), _p(self.label_key, C.c_uint32),
                                _p(self.label_val, C.c_uint32), _p(self.label_int, C.c_int64), _p(self.label_int_ok, C.c_uint8),
                                _p(self.taint_off, C.c_uint32), _p(self.taint_key, C.c_uint32), _p(self.taint_val, C.c_uint32),
                                _p(self.taint_effect, C.c_uint8))


@dataclass
class Requirements:
    key: np.ndarray
    op: np.ndarray
    val_off: np.ndarray
    val: np.ndarray
    val_int: np.ndarray
    val_int_ok: np.ndarray

    def as_struct(self) -> RequirementsStruct:
        return RequirementsStruct(len(self.key), _p(self.key, C.c_uint32), _p(self.op, C.c_uint8), _p(self.val_off, C.c_uint32),
                                  _p(self.val, C.c_uint32), _p(self.val_int, C.c_int64), _p(self.val_int_ok, C.c_uint8))


@dataclass
class FitTemplates:
    field_name_key: int
    flags: np.ndarray
    sel_off: np.ndarray
    sel_key: np.ndarray
    sel_val: np.ndarray
    term_off: np.ndarray
    term_expr_off: np.ndarray
    term_field_off: np.ndarray
    exprs: Requirements
    fields: Requirements
    tol_off: np.ndarray
    tol_key: np.ndarray
    tol_val: np.ndarray
    tol_op: np.ndarray
    tol_effect: np.ndarray

    @property
    def c(self) -> int:
        return len(self.flags)

    def as_struct(self) -> FitTemplatesStruct:
        return FitTemplatesStruct(self.c, self.field_name_key, _p(self.flags, C.c_uint8),
                                  _p(self.sel_off, C.c_uint32), _p(self.sel_key, C.c_uint32), _p(self.sel_val, C.c_uint32),
                                  _p(self.term_off, C.c_uint32), _p(self.term_expr_off, C.c_uint32), _p(self.term_field_off, C.c_uint32),
                                  self.exprs.as_struct(), self.fields.as_struct(),
                                  _p(self.tol_off, C.c_uint32), _p(self.tol_key, C.c_uint32), _p(self.tol_val, C.c_uint32),
                                  _p(self.tol_op, C.c_uint8), _p(self.tol_effect, C.c_uint8))


class _ReqBuilder:
    def __init__(self, intern: Interner, validate_strings: bool):
        self.intern, self.validate = intern, validate_strings
        self.key, self.op, self.val_off, self.val, self.val_int, self.val_int_ok = [], [], [0], [], [], []

    def add(self, key: str, op: str, values):
        code = _OPS.get(op)
        bad = code is None
        if bad:
            code = 0x7F
        if self.validate and (not label_key_ok(key) or not all(label_value_ok(v) for v in values)):
            bad = True
        self.key.append(self.intern(key))
        self.op.append(code | (OP_INVALID if bad else 0))
        for v in values:
            iv, ok = parse_int(v)
            self.val.append(self.intern(v))
            self.val_int.append(iv)
            self.val_int_ok.append(1 if ok else 0)
        self.val_off.append(len(self.val))

    def done(self) -> Requirements:
        return Requirements(_u32(self.key), _u8(self.op), _u32(self.val_off), _u32(self.val), _i64(self.val_int), _u8(self.val_int_ok))


def marshal_nodes(nodes, intern: Interner) -> NodeLabels:
    name, loff, lkey, lval, lint, lok, toff, tkey, tval, teff = [], [0], [], [], [], [], [0], [], [], []
    for nd in nodes:
        name.append(intern(nd.get("name", "")))
        for k, v in nd.get("labels", {}).items():
            iv, ok = parse_int(v)
            lkey.append(intern(k)); lval.append(intern(v)); lint.append(iv); lok.append(1 if ok else 0)
        loff.append(len(lkey))
        for (k, v, e) in nd.get("taints", []):
            tkey.append(intern(k)); tval.append(intern(v)); teff.append(intern.code(e, _EFFECTS, 4))
        toff.append(len(tkey))
    return NodeLabels(_u32(name), _u32(loff), _u32(lkey), _u32(lval), _i64(lint), _u8(lok), _u32(toff), _u32(tkey), _u32(tval), _u8(teff))


def marshal_templates(templates, intern: Interner) -> FitTemplates:
    flags, soff, skey, sval, moff, meo, mfo = [], [0], [], [], [0], [0], [0]
    ex, fl = _ReqBuilder(intern, True), _ReqBuilder(intern, False)
    ooff, okey, oval, oop, oeff = [0], [], [], [], []
    for tp in templates:
        f = 0
        sel = tp.get("node_selector") or {}
        for k, v in sel.items():
            if not label_key_ok(k) or not label_value_ok(v):
                f |= TPL_SELECTOR_INVALID
            skey.append(intern(k)); sval.append(intern(v))
        soff.append(len(skey))
        req = tp.get("required")
        if req is not None:
            f |= TPL_HAS_REQUIRED
            for term in req:
                for (k, op, vals) in term.get("expressions", []):
                    ex.add(k, op, vals)
                for (k, op, vals) in term.get("fields", []):
                    fl.add(k, op, vals)
                meo.append(len(ex.key)); mfo.append(len(fl.key))
        moff.append(len(meo) - 1)
        for (k, op, v, e) in tp.get("tolerations", []):
            okey.append(intern(k)); oval.append(intern(v))
            oop.append(intern.code(op, _TOL_OPS, 3)); oeff.append(intern.code(e, _EFFECTS, 4))
        ooff.append(len(okey))
        flags.append(f)
    return FitTemplates(intern("metadata.name"), _u8(flags), _u32(soff), _u32(skey), _u32(sval), _u32(moff), _u32(meo), _u32(mfo),
                        ex.done(), fl.done(), _u32(ooff), _u32(okey), _u32(oval), _u8(oop), _u8(oeff))


def marshal(nodes, templates):
    intern = Interner()
    return marshal_nodes(nodes, intern), marshal_templates(templates, intern)
