/*
 * bs_oracle_fit.c — CPU oracle for the fit-mask builder: checkFit (core.go:741-759) evaluated pair
 * by pair.  TEST INFRASTRUCTURE ONLY (see bs_oracle.h).
 *
 * checkFit calls two predicates of k8s.io/kubernetes v1.17.5 (go.mod:102), which is NOT vendored
 * under /root/reference; the published v1.17.5 / apimachinery v0.17.5 algorithm is restated here
 * (PARITY UNPINNED by the reference: it has no test of checkFit).  Upstream rules followed ("U6"):
 *
 *  U6.1 predicates.PodMatchNodeSelector: node == nil -> error (checkFit false, core.go:744-746);
 *       else PodMatchesNodeSelectorAndAffinityTerms:
 *         len(Spec.NodeSelector) > 0 -> labels.SelectorFromSet(NodeSelector).Matches(node.Labels);
 *         SelectorFromSet builds one `key = value` requirement per pair and, if NewRequirement
 *         rejects any pair (label key / value validation), returns the EMPTY selector, which matches
 *         every label set;
 *         Affinity.NodeAffinity != nil: Required == nil -> true; else the node must match
 *         v1helper.MatchNodeSelectorTerms(Required.NodeSelectorTerms, labels, {"metadata.name": name}).
 *  U6.2 MatchNodeSelectorTerms: terms are ORed; a term with neither matchExpressions nor matchFields
 *       is skipped; a term whose selector conversion errors is skipped; within a term
 *       matchExpressions and matchFields are ANDed.  No term (or none matching) -> false.
 *  U6.3 NodeSelectorRequirementsAsSelector -> labels.NewRequirement: In/NotIn need >= 1 value,
 *       Exists/DoesNotExist exactly 0, Gt/Lt exactly 1 that parses with ParseInt(v, 10, 64); key and
 *       values must pass label validation (the caller reports that as BS_OP_INVALID); unknown
 *       operator -> error.
 *  U6.4 Requirement.Matches: In: Has(key) && value in set; NotIn: !Has(key) || value not in set;
 *       Exists: Has; DoesNotExist: !Has; Gt/Lt: Has && ParseInt(label value) ok && lv > rv / lv < rv.
 *  U6.5 NodeSelectorRequirementsAsFieldSelector: only In / NotIn with exactly one value convert, to
 *       `field = value` / `field != value`; fields.Set.Get of a missing field is "".
 *  U6.6 predicates.PodToleratesNodeTaints: node == nil -> (false, nil); nodeInfo.Taints() error ->
 *       error (checkFit false, core.go:752-754); taints are filtered to effect NoSchedule / NoExecute;
 *       each remaining taint needs one toleration with ToleratesTaint true.
 *  U6.7 Toleration.ToleratesTaint: (Effect == "" || Effect == taint.Effect) && (Key == "" ||
 *       Key == taint.Key) && (Operator "" / Equal: Value == taint.Value; Exists: true; other: false).
 */
#include "bs_oracle.h"

/* labels.Set lookup: node.Labels is a map, keys unique */
static int label_find(const bs_node_labels* nl, uint32_t node, uint32_t key, uint32_t* at) {
  for (uint32_t e = nl->label_off[node]; e < nl->label_off[node + 1]; ++e)
    if (nl->label_key[e] == key) { *at = e; return 1; }
  return 0;
}

/* U6.3: can this requirement become a labels.Requirement? */
static int expr_converts(const bs_requirements* r, uint32_t i) {
  if (r->op[i] & BS_OP_INVALID) return 0;
  const uint32_t nv = r->val_off[i + 1] - r->val_off[i];
  switch (r->op[i]) {
    case BS_OP_IN: case BS_OP_NOT_IN: return nv > 0;
    case BS_OP_EXISTS: case BS_OP_DOES_NOT_EXIST: return nv == 0;
    case BS_OP_GT: case BS_OP_LT: return nv == 1 && r->val_int_ok[r->val_off[i]];
    default: return 0;
  }
}

/* U6.4 */
static int expr_matches(const bs_requirements* r, uint32_t i, const bs_node_labels* nl, uint32_t node) {
  uint32_t at = 0;
  const int has = label_find(nl, node, r->key[i], &at);
  int in_set = 0;
  if (has)
    for (uint32_t j = r->val_off[i]; j < r->val_off[i + 1]; ++j)
      if (r->val[j] == nl->label_val[at]) in_set = 1;
  switch (r->op[i]) {
    case BS_OP_IN: return has && in_set;
    case BS_OP_NOT_IN: return !has || !in_set;
    case BS_OP_EXISTS: return has;
    case BS_OP_DOES_NOT_EXIST: return !has;
    case BS_OP_GT: case BS_OP_LT: {
      if (!has || !nl->label_int_ok[at]) return 0;
      const int64_t lv = nl->label_int[at], rv = r->val_int[r->val_off[i]];
      return r->op[i] == BS_OP_GT ? lv > rv : lv < rv;
    }
    default: return 0;
  }
}

/* U6.2 for one term */
static int term_matches(const bs_fit_templates* tp, uint32_t t, const bs_node_labels* nl, uint32_t node) {
  const uint32_t e0 = tp->term_expr_off[t], e1 = tp->term_expr_off[t + 1];
  const uint32_t f0 = tp->term_field_off[t], f1 = tp->term_field_off[t + 1];
  if (e0 == e1 && f0 == f1) return 0;
  if (e0 != e1) {
    for (uint32_t i = e0; i < e1; ++i) if (!expr_converts(&tp->exprs, i)) return 0;   /* err -> continue */
    for (uint32_t i = e0; i < e1; ++i) if (!expr_matches(&tp->exprs, i, nl, node)) return 0;
  }
  if (f0 != f1) {
    const bs_requirements* r = &tp->fields;
    for (uint32_t i = f0; i < f1; ++i) {                                               /* U6.5 */
      if (r->op[i] != BS_OP_IN && r->op[i] != BS_OP_NOT_IN) return 0;
      if (r->val_off[i + 1] - r->val_off[i] != 1) return 0;
    }
    for (uint32_t i = f0; i < f1; ++i) {
      const uint32_t got = r->key[i] == tp->field_name_key ? nl->name[node] : 0u;      /* "" if absent */
      const int eq = got == r->val[r->val_off[i]];
      if (r->op[i] == BS_OP_IN ? !eq : eq) return 0;
    }
  }
  return 1;
}

/* U6.1 */
static int pod_matches_node_selector(const bs_fit_templates* tp, uint32_t c, const bs_node_labels* nl, uint32_t node) {
  if (tp->sel_off[c + 1] > tp->sel_off[c] && !(tp->flags[c] & BS_TPL_SELECTOR_INVALID)) {
    for (uint32_t i = tp->sel_off[c]; i < tp->sel_off[c + 1]; ++i) {
      uint32_t at = 0;
      if (!label_find(nl, node, tp->sel_key[i], &at)) return 0;
      if (nl->label_val[at] != tp->sel_val[i]) return 0;
    }
  }
  if (!(tp->flags[c] & BS_TPL_HAS_REQUIRED)) return 1;
  for (uint32_t t = tp->term_off[c]; t < tp->term_off[c + 1]; ++t)
    if (term_matches(tp, t, nl, node)) return 1;
  return 0;
}

/* U6.7 */
static int tolerates(const bs_fit_templates* tp, uint32_t o, uint32_t key, uint32_t val, uint8_t effect) {
  if (tp->tol_effect[o] != BS_EFFECT_NONE && tp->tol_effect[o] != effect) return 0;
  if (tp->tol_key[o] != 0 && tp->tol_key[o] != key) return 0;
  switch (tp->tol_op[o]) {
    case BS_TOL_OP_DEFAULT: case BS_TOL_OP_EQUAL: return tp->tol_val[o] == val;
    case BS_TOL_OP_EXISTS: return 1;
    default: return 0;
  }
}

/* U6.6 */
static int pod_tolerates_node_taints(const bs_fit_templates* tp, uint32_t c, const bs_node_labels* nl, uint32_t node) {
  for (uint32_t t = nl->taint_off[node]; t < nl->taint_off[node + 1]; ++t) {
    const uint8_t eff = nl->taint_effect[t];
    if (eff != BS_EFFECT_NO_SCHEDULE && eff != BS_EFFECT_NO_EXECUTE) continue;
    int ok = 0;
    for (uint32_t o = tp->tol_off[c]; o < tp->tol_off[c + 1] && !ok; ++o)
      ok = tolerates(tp, o, nl->taint_key[t], nl->taint_val[t], eff);
    if (!ok) return 0;
  }
  return 1;
}

/* checkFit, core.go:741-759 */
int orc_check_fit(const bs_node_labels* nl, const uint8_t* node_flags, const bs_fit_templates* tp, uint32_t c, uint32_t node) {
  if (node_flags[node] & (BS_NODE_NIL | BS_NODE_NO_NODE)) return 0;   /* :744 err / U6.6 node == nil */
  if (node_flags[node] & BS_NODE_TAINT_ERR) return 0;                 /* :752-754 */
  int fails = 0;                                                       /* len(predicateFails) */
  if (!pod_matches_node_selector(tp, c, nl, node)) fails++;
  if (!pod_tolerates_node_taints(tp, c, nl, node)) fails++;
  return fails == 0;
}

void orc_fit_build(const bs_node_labels* nl, const uint8_t* node_flags, const bs_fit_templates* tp, uint32_t* fit_bits) {
  const uint32_t words = (nl->n + 31u) / 32u;
  for (uint32_t c = 0; c < tp->c; ++c) {
    uint32_t* row = fit_bits + (size_t)c * words;
    for (uint32_t w = 0; w < words; ++w) row[w] = 0;
    for (uint32_t n = 0; n < nl->n; ++n)
      if (orc_check_fit(nl, node_flags, tp, c, n)) row[n >> 5] |= 1u << (n & 31);
  }
}
