// bs_fit.hpp — fit-mask builder: checkFit (core.go:741-759) for every (pod-template class, node).
//
// checkFit = PodMatchNodeSelector && PodToleratesNodeTaints of k8s.io/kubernetes v1.17.5 (not vendored;
// the upstream rules followed are listed in DESIGN.md, "Fit-mask builder", U6.1-U6.7).  Strings arrive as interned ids, so the
// work is a join of the templates' keys against every node's label set:
//
//   k_fit_cols   gather: only the K label keys some template mentions matter, so the node labels (CSR,
//                one ragged row per node) are re-laid as K dense columns [K][Npad] of (value id,
//                present/int_ok bits, parsed integer).  One thread per node, binary search of each of its
//                label keys in the sorted key list.
//   k_fit_match  lane = node, block = 256 consecutive nodes, blockIdx.y strides over classes.  The
//                template is wave-uniform (scalar loads, uniform branches); every requirement is one
//                coalesced column read.  Taints are a ragged per-node loop against the class's
//                tolerations.  A wave ballots its 64 verdicts into two mask words.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsched.h"

namespace bs {

struct FitCols {
  uint32_t K, Npad;
  const uint32_t* keys;   // [K] sorted referenced key ids
  uint32_t* val;          // [K][Npad]
  uint8_t* flag;          // [K][Npad] bit0 present, bit1 int_ok
  int64_t* ival;          // [K][Npad]
};

struct FitNodesDev {
  uint32_t n;
  const uint32_t* name;
  const uint32_t *label_off, *label_key, *label_val;
  const int64_t* label_int;
  const uint8_t* label_int_ok;
  const uint32_t *taint_off, *taint_key, *taint_val;
  const uint8_t* taint_effect;
  const uint8_t* nflags;  // BS_NODE_* of the loaded snapshot
};

struct FitReqDev {
  const uint32_t* key;    // exprs: column index (BS_FIT_NO_COL never happens: every expr key is a column); fields: raw id
  const uint8_t* op;
  const uint32_t* val_off;
  const uint32_t* val;
  const int64_t* val_int;
  const uint8_t* val_int_ok;
};

struct FitTplDev {
  uint32_t c, field_name_key;
  const uint8_t* flags;
  const uint32_t *sel_off, *sel_col, *sel_val;
  const uint32_t *term_off, *term_expr_off, *term_field_off;
  FitReqDev ex, fl;
  const uint32_t *tol_off, *tol_key, *tol_val;
  const uint8_t *tol_op, *tol_effect;
};

__global__ __launch_bounds__(256) void k_fit_cols(FitNodesDev nd, FitCols cols) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  if (n >= nd.n || cols.K == 0) return;
  for (uint32_t e = nd.label_off[n], e1 = nd.label_off[n + 1]; e < e1; ++e) {
    const uint32_t key = nd.label_key[e];
    uint32_t lo = 0, hi = cols.K;                 // first index with keys[i] >= key
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (cols.keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    if (lo < cols.K && cols.keys[lo] == key) {
      const size_t at = (size_t)lo * cols.Npad + n;
      cols.val[at] = nd.label_val[e];
      cols.flag[at] = (uint8_t)(1u | (nd.label_int_ok[e] ? 2u : 0u));
      cols.ival[at] = nd.label_int[e];
    }
  }
}

// One v1.NodeSelectorRequirement as labels.Requirement (selector.go NewRequirement + Matches).
// `valid` is wave-uniform (a property of the requirement), `match` is per node.
__device__ __forceinline__ void fit_expr(const FitReqDev& r, uint32_t i, const FitCols& cols, uint32_t n, bool& valid, bool& match) {
  const uint8_t opb = r.op[i];
  const uint32_t op = opb & 0x7Fu;
  const uint32_t v0 = r.val_off[i], nv = r.val_off[i + 1] - v0;
  valid = !(opb & BS_OP_INVALID);
  const size_t at = (size_t)r.key[i] * cols.Npad + n;
  const uint32_t fl = cols.flag[at];
  const bool pres = fl & 1u;
  const uint32_t v = cols.val[at];
  match = false;
  switch (op) {
    case BS_OP_IN:
    case BS_OP_NOT_IN: {
      valid = valid && nv > 0;
      bool has = false;
      for (uint32_t j = 0; j < nv; ++j) has |= r.val[v0 + j] == v;
      match = op == BS_OP_IN ? (pres && has) : (!pres || !has);
      break;
    }
    case BS_OP_EXISTS: valid = valid && nv == 0; match = pres; break;
    case BS_OP_DOES_NOT_EXIST: valid = valid && nv == 0; match = !pres; break;
    case BS_OP_GT:
    case BS_OP_LT: {
      valid = valid && nv == 1 && r.val_int_ok[v0];
      if (valid) {
        const int64_t rv = r.val_int[v0], lv = cols.ival[at];
        match = pres && (fl & 2u) && (op == BS_OP_GT ? lv > rv : lv < rv);
      }
      break;
    }
    default: valid = false;
  }
}

// One matchFields requirement as a field selector (NodeSelectorRequirementsAsFieldSelector): only
// In / NotIn with exactly one value convert; the node's field set is {"metadata.name": name}, any
// other key reads as "".
__device__ __forceinline__ void fit_field(const FitReqDev& r, uint32_t i, uint32_t field_name_key, uint32_t name, bool& valid, bool& match) {
  const uint8_t opb = r.op[i];
  const uint32_t op = opb & 0x7Fu;
  const uint32_t v0 = r.val_off[i], nv = r.val_off[i + 1] - v0;
  valid = !(opb & BS_OP_INVALID) && (op == BS_OP_IN || op == BS_OP_NOT_IN) && nv == 1;
  match = false;
  if (valid) {
    const uint32_t lhs = r.key[i] == field_name_key ? name : 0u;
    const bool eq = lhs == r.val[v0];
    match = op == BS_OP_IN ? eq : !eq;
  }
}

__global__ __launch_bounds__(256) void k_fit_match(FitNodesDev nd, FitTplDev tp, FitCols cols, uint32_t* __restrict__ fit, uint32_t fit_words) {
  const uint32_t n = blockIdx.x * 256u + threadIdx.x;
  const bool in = n < nd.n;
  const uint32_t nn = in ? n : 0u;
  const bool node_ok = in && !(nd.nflags[nn] & (BS_NODE_NIL | BS_NODE_NO_NODE | BS_NODE_TAINT_ERR));
  const uint32_t name = in ? nd.name[nn] : 0u;
  const uint32_t t0 = in ? nd.taint_off[nn] : 0u, t1 = in ? nd.taint_off[nn + 1] : 0u;
  const int lane = threadIdx.x & 63;
  const uint32_t w32 = (blockIdx.x * 256u + (threadIdx.x & ~63u)) >> 5;   // first mask word of this wave

  for (uint32_t c = blockIdx.y; c < tp.c; c += gridDim.y) {
    bool ok = node_ok;
    const uint32_t tf = tp.flags[c];
    // ---- Spec.NodeSelector (labels.SelectorFromSet: every pair key == value)
    if (!(tf & BS_TPL_SELECTOR_INVALID)) {
      for (uint32_t i = tp.sel_off[c], i1 = tp.sel_off[c + 1]; i < i1; ++i) {
        const size_t at = (size_t)tp.sel_col[i] * cols.Npad + nn;
        ok = ok && (cols.flag[at] & 1u) && cols.val[at] == tp.sel_val[i];
      }
    }
    // ---- required node affinity: terms ORed, requirements ANDed
    if (tf & BS_TPL_HAS_REQUIRED) {
      bool any = false;
      for (uint32_t t = tp.term_off[c], te = tp.term_off[c + 1]; t < te; ++t) {
        const uint32_t e0 = tp.term_expr_off[t], e1 = tp.term_expr_off[t + 1];
        const uint32_t f0 = tp.term_field_off[t], f1 = tp.term_field_off[t + 1];
        if (e0 == e1 && f0 == f1) continue;       // nil or empty term selects no objects
        bool m = true;
        for (uint32_t i = e0; i < e1; ++i) {
          bool valid, match;
          fit_expr(tp.ex, i, cols, nn, valid, match);
          m = m && valid && match;
        }
        for (uint32_t i = f0; i < f1; ++i) {
          bool valid, match;
          fit_field(tp.fl, i, tp.field_name_key, name, valid, match);
          m = m && valid && match;
        }
        any = any || m;
      }
      ok = ok && any;
    }
    // ---- taints with effect NoSchedule / NoExecute must each be tolerated by some toleration
    if (ok) {
      const uint32_t o0 = tp.tol_off[c], o1 = tp.tol_off[c + 1];
      for (uint32_t t = t0; t < t1 && ok; ++t) {
        const uint32_t eff = nd.taint_effect[t];
        if (eff != BS_EFFECT_NO_SCHEDULE && eff != BS_EFFECT_NO_EXECUTE) continue;
        const uint32_t tk = nd.taint_key[t], tv = nd.taint_val[t];
        bool tolerated = false;
        for (uint32_t o = o0; o < o1; ++o) {
          const uint32_t oe = tp.tol_effect[o], ok_ = tp.tol_key[o], op = tp.tol_op[o];
          if (oe != BS_EFFECT_NONE && oe != eff) continue;
          if (ok_ != 0u && ok_ != tk) continue;
          if (op == BS_TOL_OP_EXISTS) tolerated = true;
          else if (op == BS_TOL_OP_DEFAULT || op == BS_TOL_OP_EQUAL) tolerated = tolerated || tp.tol_val[o] == tv;
        }
        ok = tolerated;
      }
    }
    const unsigned long long bal = __ballot(ok);
    if (lane == 0) {
      uint32_t* row = fit + (size_t)c * fit_words;
      if (w32 < fit_words) row[w32] = (uint32_t)bal;
      if (w32 + 1 < fit_words) row[w32 + 1] = (uint32_t)(bal >> 32);
    }
  }
}

}  // namespace bs
