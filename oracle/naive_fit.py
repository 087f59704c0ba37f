"""Independent string-level restatement of checkFit (core.go:741-759) — TEST INFRASTRUCTURE ONLY.

Works on the object form (dicts and strings, see batch-scheduler_amd/fitspec.py) exactly as the
k8s.io/kubernetes v1.17.5 predicates work on *v1.Pod / *v1.Node, with its own string validation and
integer parsing, so it checks both the interning marshaller and the C oracle (bs_oracle_fit.c), which
only ever see ids.  Rule numbers (U6.x) refer to the list at the top of bs_oracle_fit.c.
"""
from __future__ import annotations

import string

_ALNUM = set(string.ascii_letters + string.digits)
_NAME_CHARS = _ALNUM | set("-_.")
_LOWER_ALNUM = set(string.ascii_lowercase + string.digits)


def _name_part_ok(s: str) -> bool:
    return 0 < len(s) <= 63 and s[0] in _ALNUM and s[-1] in _ALNUM and all(ch in _NAME_CHARS for ch in s)


def _dns_label_ok(s: str) -> bool:
    return len(s) > 0 and s[0] in _LOWER_ALNUM and s[-1] in _LOWER_ALNUM and all(ch in _LOWER_ALNUM or ch == "-" for ch in s)


def qualified_name_ok(key: str) -> bool:
    parts = key.split("/")
    if len(parts) > 2:
        return False
    if len(parts) == 2:
        prefix = parts[0]
        if len(prefix) == 0 or len(prefix) > 253 or not all(_dns_label_ok(x) for x in prefix.split(".")):
            return False
    return _name_part_ok(parts[-1])


def label_value_ok(v: str) -> bool:
    return v == "" or _name_part_ok(v)


def go_parse_int(s: str):
    """strconv.ParseInt(s, 10, 64) -> value or None."""
    body = s[1:] if s[:1] in ("+", "-") else s
    if body == "" or any(ch not in string.digits for ch in body):
        return None
    v = int(s)
    return v if -(2 ** 63) <= v < 2 ** 63 else None


def _requirement(key, op, values):
    """labels.NewRequirement (U6.3): returns a matcher(labels) or None when the conversion errors."""
    if not qualified_name_ok(key):
        return None
    if op in ("In", "NotIn"):
        if len(values) == 0:
            return None
    elif op in ("Exists", "DoesNotExist"):
        if len(values) != 0:
            return None
    elif op in ("Gt", "Lt"):
        if len(values) != 1 or go_parse_int(values[0]) is None:
            return None
    else:
        return None
    if not all(label_value_ok(v) for v in values):
        return None

    def matches(labels):                                            # U6.4
        if op == "In":
            return key in labels and labels[key] in values
        if op == "NotIn":
            return key not in labels or labels[key] not in values
        if op == "Exists":
            return key in labels
        if op == "DoesNotExist":
            return key not in labels
        if key not in labels:
            return False
        lv = go_parse_int(labels[key])
        if lv is None:
            return False
        rv = go_parse_int(values[0])
        return lv > rv if op == "Gt" else lv < rv
    return matches


def _field_requirement(key, op, values):
    """NodeSelectorRequirementsAsFieldSelector (U6.5)."""
    if op not in ("In", "NotIn") or len(values) != 1:
        return None
    if op == "In":
        return lambda fields: fields.get(key, "") == values[0]
    return lambda fields: fields.get(key, "") != values[0]


def match_node_selector_terms(terms, labels, fields) -> bool:       # U6.2
    for term in terms:
        exprs, flds = term.get("expressions", []), term.get("fields", [])
        if len(exprs) == 0 and len(flds) == 0:
            continue
        if exprs:
            ms = [_requirement(*e) for e in exprs]
            if any(m is None for m in ms) or not all(m(labels) for m in ms):
                continue
        if flds:
            ms = [_field_requirement(*f) for f in flds]
            if any(m is None for m in ms) or not all(m(fields) for m in ms):
                continue
        return True
    return False


def pod_matches_node_selector(tpl, node) -> bool:                  # U6.1
    labels = node.get("labels", {})
    sel = tpl.get("node_selector") or {}
    if len(sel) > 0:
        if all(qualified_name_ok(k) and label_value_ok(v) for k, v in sel.items()):   # else: empty selector
            for k, v in sel.items():
                if k not in labels or labels[k] != v:
                    return False
    req = tpl.get("required")
    if req is None:
        return True
    return match_node_selector_terms(req, labels, {"metadata.name": node.get("name", "")})


def tolerates(tol, taint) -> bool:                                 # U6.7
    tkey, top, tval, teff = tol
    key, val, eff = taint
    if len(teff) > 0 and teff != eff:
        return False
    if len(tkey) > 0 and tkey != key:
        return False
    if top in ("", "Equal"):
        return tval == val
    return top == "Exists"


def pod_tolerates_node_taints(tpl, node) -> bool:                  # U6.6
    for taint in node.get("taints", []):
        if taint[2] not in ("NoSchedule", "NoExecute"):
            continue
        if not any(tolerates(tol, taint) for tol in tpl.get("tolerations", [])):
            return False
    return True


def check_fit(tpl, node, node_flags: int = 0) -> bool:
    """core.go:741-759; node_flags are BS_NODE_* (nil entry, nil Node(), Taints() error)."""
    if node_flags & 0x0B:
        return False
    fails = 0
    if not pod_matches_node_selector(tpl, node):
        fails += 1
    if not pod_tolerates_node_taints(tpl, node):
        fails += 1
    return fails == 0
