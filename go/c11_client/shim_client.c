/* shim_client.c — the call sequence of go/pkg/scheduler/core/bsched_cgo.go + bsched_batch.go, in plain C11.
 *
 * The image this library is developed in has no Go toolchain, so the cgo shim cannot be compiled here.  What cgo does
 * with include/bsched.h is: parse it as C, build the argument structs field by field, call the entry points.  This file
 * does exactly that, in the shim's order, through the same header and the same shared library:
 *
 *   newGPUCore      bs_abi_version, bs_create                                   (bsched_cgo.go: newGPUCore)
 *   loadSnapshot    bs_nodes_load_flat, bs_fit_load                             (bsched_cgo.go: loadSnapshot, core.go:437,567,597)
 *   loadGroups      bs_groups_load_flat                                         (bsched_batch.go: loadGroups)
 *   runBatch        bs_pods_load_flat, bs_batch_run(BS_STAGE_ALL), bs_filter_rows_count, bs_batch_read_flat   (bsched_batch.go: runBatch)
 *   clusterFits     bs_cluster_fits for the first pods of the queue              (bsched_cgo.go: clusterFits, core.go:595-632)
 *   next cycle      bs_groups_apply (patchGroups), bs_pods_apply_flat, bs_batch_run(| BS_BATCH_HOST_RESULTS), bs_batch_map
 *   Filter on       bs_batch_run(| BS_BATCH_FILTER_DENY), bs_batch_read_flat, bs_filter_deny_stats, bs_speculation_stats
 *   close           bs_destroy
 * The struct-taking entry points are reached through their *_flat forms only, exactly as the Go files do (cgo pointer rule: no
 * Go-allocated struct of Go pointers crosses by pointer) — so the forms the shim binds are compiled, linked and run here.
 *
 * Built with  gcc -std=c11 -Wall -Wextra -Wpedantic -Werror  (tests/test_c11_client.py: header is valid C11, every symbol
 * the shim binds resolves against libbsched.so); run on a GPU box against a scene file written by the test, whose results
 * have to equal the ctypes binding's and the CPU oracle's on the same scene.
 *
 * usage: shim_client <scene.bin> <result.bin>
 * scene.bin   : u32 magic 0x42534331, L, N, C, G, P, fit_words | nodes: alloc[L][N] req[L][N] apres[N] rpres[N] flags[N] |
 *               fit[C][fit_words] | groups: mm sc matched [G] u32, flags[G] u8, cls[G] u32, minres[L][G] i64, mrpres[G] u32,
 *               occ[G] u64 | pods: group[P] i32, req[L][P] i64, pres[P] u32, cls[P] u32, owner[P] u64, flags[P] u8
 * result.bin  : cycle 1 then cycle 2, each: pf_code[P'] pf_first_k[P'] pf_leader[P'] fl_code[P'] fl_feasible[P'] admit[G] ready[G];
 *               then fits[min(P,8)] u8 + first_k[min(P,8)] u32 of the 1:1 calls
 */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bsched.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != BS_OK) {                                                                      \
      fprintf(stderr, "%s: %s (%s)\n", #call, bs_strerror(rc_), ctx ? bs_last_error(ctx) : ""); \
      return 2;                                                                              \
    }                                                                                        \
  } while (0)

static void* rd(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || (bytes && fread(p, 1, bytes, f) != bytes)) {
    fprintf(stderr, "scene file too short\n");
    exit(3);
  }
  return p;
}

static void wr(FILE* f, const void* p, size_t bytes) {
  if (bytes && fwrite(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "cannot write the result file\n");
    exit(3);
  }
}

int main(int argc, char** argv) {
  bs_ctx* ctx = NULL;
  if (argc != 3) {
    fprintf(stderr, "usage: %s scene.bin result.bin\n", argv[0]);
    return 1;
  }
  if (bs_abi_version() != BS_ABI_VERSION) {
    fprintf(stderr, "header ABI %u, library ABI %u\n", BS_ABI_VERSION, bs_abi_version());
    return 1;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  uint32_t hdr[7];
  if (fread(hdr, 4, 7, f) != 7 || hdr[0] != 0x42534331u) return 1;
  const uint32_t L = hdr[1], N = hdr[2], C = hdr[3], G = hdr[4], P = hdr[5], FW = hdr[6];

  /* ---- newGPUCore */
  bs_config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = BS_ABI_VERSION;
  cfg.device = 0;
  cfg.scalar_lanes = L - 4;
  cfg.eph_gate = 1;
  CHECK(bs_create(&cfg, &ctx));

  /* ---- loadSnapshot */
  bs_nodes_soa nodes;
  nodes.n = N;
  nodes.allocatable = rd(f, (size_t)L * N * 8);
  nodes.requested = rd(f, (size_t)L * N * 8);
  nodes.allocatable_present = rd(f, (size_t)N * 4);
  nodes.requested_present = rd(f, (size_t)N * 4);
  nodes.flags = rd(f, N);
  const uint32_t* fit = rd(f, (size_t)C * FW * 4);
  CHECK(bs_nodes_load_flat(ctx, nodes.n, nodes.allocatable, nodes.requested, nodes.allocatable_present, nodes.requested_present, nodes.flags));
  CHECK(bs_fit_load(ctx, C, fit));
  uint32_t n_back = 0;
  CHECK(bs_nodes_count(ctx, &n_back));
  if (n_back != N) return 4;

  /* ---- loadGroups */
  bs_groups_soa groups;
  groups.g = G;
  groups.min_member = rd(f, (size_t)G * 4);
  groups.status_scheduled = rd(f, (size_t)G * 4);
  groups.matched = rd(f, (size_t)G * 4);
  groups.flags = rd(f, G);
  groups.cls = rd(f, (size_t)G * 4);
  groups.min_resources = rd(f, (size_t)L * G * 8);
  groups.min_resources_present = rd(f, (size_t)G * 4);
  groups.occupied_by = rd(f, (size_t)G * 8);
  CHECK(bs_groups_load_flat(ctx, groups.g, groups.min_member, groups.status_scheduled, groups.matched, groups.flags, groups.cls, groups.min_resources,
                            groups.min_resources_present, groups.occupied_by));

  /* ---- runBatch */
  bs_pods_soa pods;
  pods.p = P;
  int32_t* pgroup = rd(f, (size_t)P * 4);
  int64_t* preq = rd(f, (size_t)L * P * 8);
  uint32_t* ppres = rd(f, (size_t)P * 4);
  uint32_t* pcls = rd(f, (size_t)P * 4);
  uint64_t* powner = rd(f, (size_t)P * 8);
  uint8_t* pflags = rd(f, P);
  fclose(f);
  pods.group = pgroup; pods.req = preq; pods.req_present = ppres; pods.cls = pcls; pods.owner = powner; pods.flags = pflags;
  CHECK(bs_pods_load_flat(ctx, pods.p, pods.group, pods.req, pods.req_present, pods.cls, pods.owner, pods.flags));
  CHECK(bs_batch_run(ctx, BS_STAGE_ALL));
  uint32_t rows_needed = 0;
  CHECK(bs_filter_rows_count(ctx, &rows_needed));
  const uint32_t W = (N + 63) / 64, rows_cap = rows_needed + 1;
  uint8_t* pf_code = calloc(P + 1, 1);
  uint32_t* pf_first_k = calloc(P + 1, 4);
  int32_t* pf_leader = calloc(P + 1, 4);
  uint8_t* fl_code = calloc(P + 1, 1);
  uint32_t* fl_feasible = calloc(P + 1, 4);
  uint32_t* fl_slot = calloc(P + 1, 4);
  uint64_t* rows = calloc((size_t)W * rows_cap + 1, 8);
  uint32_t* admit = calloc(G + 1, 4);
  uint8_t* ready = calloc(G + 1, 1);
  uint32_t rows_n = 0;
  CHECK(bs_batch_read_flat(ctx, pf_code, pf_first_k, pf_leader, fl_code, fl_feasible, NULL, admit, ready, fl_slot, rows, NULL, rows_cap, &rows_n));
  /* batchResult.filterPasses: the Filter answer is a bit test in the rows — it has to agree with the per-pod feasible count */
  for (uint32_t i = 0; i < P; ++i) {
    if (fl_code[i] != BS_FL_EVALUATED) continue;
    uint32_t feas = 0;
    for (uint32_t k = 0; k < N; ++k) feas += (uint32_t)((rows[(size_t)(k >> 6) * rows_cap + fl_slot[i]] >> (k & 63)) & 1u);
    if (feas != fl_feasible[i]) {
      fprintf(stderr, "pod %u: rows say %u feasible nodes, fl_feasible says %u\n", i, feas, fl_feasible[i]);
      return 5;
    }
  }
  FILE* o = fopen(argv[2], "wb");
  if (!o) return 1;
  wr(o, pf_code, P); wr(o, pf_first_k, (size_t)P * 4); wr(o, pf_leader, (size_t)P * 4); wr(o, fl_code, P); wr(o, fl_feasible, (size_t)P * 4);
  wr(o, admit, (size_t)G * 4); wr(o, ready, G);

  /* ---- clusterFits (the 1:1 drop-in for compareClusterResourceAndRequire): request = pod's own, percent 1 */
  const uint32_t nq = P < 8 ? P : 8;
  uint8_t fits[8] = {0};
  uint32_t fk[8] = {0};
  for (uint32_t i = 0; i < nq; ++i) {
    int64_t req[BS_MAX_LANES] = {0};
    for (uint32_t j = 0; j < L; ++j) req[j] = preq[(size_t)j * P + i];
    CHECK(bs_cluster_fits(ctx, pcls[i], 1.0f, req, ppres[i], &fits[i], &fk[i]));
  }

  /* ---- next cycle: patchGroups (two groups, same values), three pods leave the head of the queue, three new ones (clones of
   * pods 3, 4, 5) arrive at its tail; latency mode; results read in place */
  bs_group_delta gd[2];
  const uint32_t ngd = G < 2 ? G : 2;
  for (uint32_t k = 0; k < ngd; ++k) {
    gd[k].index = k; gd[k].matched = groups.matched[k]; gd[k].status_scheduled = groups.status_scheduled[k]; gd[k].flags = groups.flags[k];
  }
  CHECK(bs_groups_apply(ctx, gd, ngd));
  const uint32_t nmove = P >= 6 ? 3 : 0;
  uint32_t remove[3] = {0, 1, 2};
  int32_t ig[3]; int64_t ireq[3 * BS_MAX_LANES]; uint32_t ipres[3], icls[3]; uint64_t iown[3]; uint8_t ifl[3];
  for (uint32_t k = 0; k < nmove; ++k) {
    const uint32_t s = 3 + k;
    ig[k] = pgroup[s]; ipres[k] = ppres[s]; icls[k] = pcls[s]; iown[k] = powner[s]; ifl[k] = pflags[s];
    for (uint32_t j = 0; j < L; ++j) ireq[(size_t)j * nmove + k] = preq[(size_t)j * P + s];
  }
  CHECK(bs_pods_apply_flat(ctx, nmove, remove, 0, NULL, NULL, nmove, ig, ireq, ipres, icls, iown, ifl, NULL /* append */));
  uint32_t p2 = 0;
  CHECK(bs_pods_count(ctx, &p2));
  if (p2 != P) return 6;
  bs_batch_view v;
  for (int attempt = 0;; ++attempt) {   /* runCycle's loop: BS_ERR_RETRY (ABI v7) = results void, cause repaired, run the batch again */
    CHECK(bs_batch_run(ctx, BS_STAGE_ALL | BS_BATCH_HOST_RESULTS));
    const int mrc = bs_batch_map(ctx, &v);
    if (mrc == BS_ERR_RETRY && attempt < 2) continue;
    CHECK(mrc);
    break;
  }
  if (v.p != P || v.g != G) return 7;
  wr(o, v.pf_code, P); wr(o, v.pf_first_k, (size_t)P * 4); wr(o, v.pf_leader, (size_t)P * 4); wr(o, v.fl_code, P); wr(o, v.fl_feasible, (size_t)P * 4);
  wr(o, v.group_admit, (size_t)G * 4); wr(o, v.group_ready, G);
  wr(o, fits, nq); wr(o, fk, (size_t)nq * 4);

  /* ---- a Filter-on deployment (runBatch in bsched_batch.go): BS_BATCH_FILTER_DENY replays the deny entry a failing Filter writes
   * (core.go:183-185) inside the batch, on the device; results copied out through the flat read */
  CHECK(bs_batch_run(ctx, BS_STAGE_ALL | BS_BATCH_FILTER_DENY));
  CHECK(bs_batch_read_flat(ctx, pf_code, pf_first_k, pf_leader, fl_code, fl_feasible, NULL, admit, ready, NULL, NULL, NULL, 0, NULL));
  wr(o, pf_code, P); wr(o, pf_first_k, (size_t)P * 4); wr(o, pf_leader, (size_t)P * 4); wr(o, fl_code, P); wr(o, fl_feasible, (size_t)P * 4);
  wr(o, admit, (size_t)G * 4); wr(o, ready, G);
  uint64_t reruns = 0, guessed = 0, missed = 0;
  CHECK(bs_filter_deny_stats(ctx, &reruns));
  CHECK(bs_speculation_stats(ctx, &guessed, &missed));
  if (missed > guessed) return 8;
  fclose(o);
  CHECK(bs_destroy(ctx));
  printf("shim_client: ok (%" PRIu32 " pods, %" PRIu32 " groups, %" PRIu32 " nodes, %" PRIu32 " filter rows)\n", P, G, N, rows_n);
  return 0;
}
