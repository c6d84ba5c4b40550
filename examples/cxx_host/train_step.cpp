// A training step driven from C++ through the C-ABI alone (include/reactranker_hip.h): no Python, no torch.
//
//   train_step <shard file> <step index> <weights file> <dropout p> <seed>
//
// Reads one packed step from a shard file (layout: reactranker_amd/shards.py), uploads it with ONE copy, rebuilds the
// feature arrays that do not travel (reactant rows = gathers of the distinct reactants' rows; f_bonds = f_atoms[b2a] ++
// bond columns), then runs  rr_reaction_forward -> ListMLE (rr_listmle_step_f32: loss + gradient in one launch) -> rr_reaction_backward -> the gradient
// all-reduce of a data-parallel job (rr_allreduce_f32; one rank here)  and prints the loss and one checksum per gradient
// tensor as JSON.  tests/test_gpu_cxx_host.py compares them with the Python modules
// on the same weights, step and dropout stream.
//
// Weights file: int32 header {H, depth, diff_depth, n_ffn, F, task_num, head}, then float32 tensors in the order
// enc W_i w,b  W_h w,b  W_o w,b   diff W_i w,b  W_h w,b  W_o w,b   ffn (w,b) x n_ffn   (absent layers: nothing).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <string>
#include <vector>

#include "reactranker_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define RR_OK_(x) do { int s_ = (x); if (s_ != 0) { fprintf(stderr, "%s failed: %s (%d) at line %d\n", #x, rr_strerror(s_), s_, __LINE__); exit(3); } } while (0)

static const char* TABLES[] = {"a2b", "b2a", "b2revb", "a2a", "a_scope", "a2b_rev_t", "b2t", "a2a_t", "npad", "atom2mol", "b2b_t", "npad_b"};
static std::vector<std::string> key_names() {                 // reactranker_amd/shards.py: _KEYS
  std::vector<std::string> k;
  for (const char* side : {"p", "r", "u"}) {
    k.push_back(std::string(side) + ".f_atoms");
    k.push_back(std::string(side) + ".fbond");
    for (const char* t : TABLES) k.push_back(std::string(side) + "." + t);
  }
  for (const char* t : {"amap", "amap_t", "bmap", "bmap_t", "scope", "targets", "add"}) k.push_back(t);
  return k;
}

struct Sec { int64_t rows, cols, off; };

template <class T> static T* dmalloc(size_t n) { void* p; HIP_OK(hipMalloc(&p, (n ? n : 1) * sizeof(T))); return static_cast<T*>(p); }

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s shard step weights dropout seed\n", argv[0]); return 1; }
  const int step_i = atoi(argv[2]);
  const float drop_p = static_cast<float>(atof(argv[4]));
  const uint64_t seed = strtoull(argv[5], nullptr, 10);

  // ---- shard file
  int fd = open(argv[1], O_RDONLY);
  if (fd < 0) { perror("open shard"); return 1; }
  struct stat sb; fstat(fd, &sb);
  const uint8_t* mm = static_cast<const uint8_t*>(mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0));
  if (memcmp(mm, "RRSHARD1", 8) != 0) { fprintf(stderr, "not a shard file\n"); return 1; }
  uint32_t ver, nsteps, afd, bfd; uint64_t idx_off;
  memcpy(&ver, mm + 8, 4); memcpy(&nsteps, mm + 12, 4); memcpy(&idx_off, mm + 16, 8); memcpy(&afd, mm + 24, 4); memcpy(&bfd, mm + 28, 4);
  if (ver != 1 || step_i < 0 || step_i >= static_cast<int>(nsteps)) { fprintf(stderr, "bad version / step\n"); return 1; }
  const int64_t* ix = reinterpret_cast<const int64_t*>(mm + idx_off) + 8 * step_i;
  const int64_t boff = ix[0], bbytes = ix[1], Q = ix[2], M = ix[3], K = ix[6], has_u = ix[7];
  const uint8_t* blob = mm + boff;
  int64_t nsec; memcpy(&nsec, blob, 8);
  const std::vector<std::string> names = key_names();
  std::map<std::string, Sec> toc;
  for (int64_t i = 0; i < nsec; ++i) {
    int64_t e[4]; memcpy(e, blob + 16 + 32 * i, 32);
    toc[names[e[0]]] = Sec{e[1], e[2], e[3]};
  }
  // ---- ONE upload; every array is a typed view of the device blob
  uint8_t* dblob = dmalloc<uint8_t>(bbytes);
  HIP_OK(hipMemcpy(dblob, blob, bbytes, hipMemcpyHostToDevice));
  auto F32 = [&](const std::string& k) -> float* { return toc.count(k) ? reinterpret_cast<float*>(dblob + toc[k].off) : nullptr; };
  auto I32 = [&](const std::string& k) -> int32_t* { return toc.count(k) ? reinterpret_cast<int32_t*>(dblob + toc[k].off) : nullptr; };
  hipStream_t st; HIP_OK(hipStreamCreate(&st));

  auto graph = [&](const std::string& s, int64_t n_mols, float* f_atoms, int64_t ld_fa, float* fbond, int64_t ld_fbb, bool want_fb) {
    rr_graph g; memset(&g, 0, sizeof(g));
    g.nA = toc[s + ".a2b"].rows; g.nB = toc[s + ".b2a"].rows; g.M = n_mols; g.K = static_cast<int>(K);
    g.Kb = static_cast<int>(toc[s + ".b2b_t"].cols);
    g.f_atoms = f_atoms; g.ld_fa = ld_fa;
    g.a2b = I32(s + ".a2b"); g.b2a = I32(s + ".b2a"); g.b2revb = I32(s + ".b2revb"); g.a2a = I32(s + ".a2a");
    g.a_scope = I32(s + ".a_scope"); g.b2t = I32(s + ".b2t"); g.a2a_t = I32(s + ".a2a_t"); g.atom2mol = I32(s + ".atom2mol");
    g.b2b_t = I32(s + ".b2b_t"); g.npad = F32(s + ".npad"); g.npad_b = F32(s + ".npad_b");
    if (want_fb) {                                              // f_bonds = [f_atoms[b2a] | bond columns | 0]  (featurization.py:198-199)
      const int64_t ld = (afd + bfd + 3) / 4 * 4;
      float* fb = dmalloc<float>(g.nB * ld);
      RR_OK_(rr_build_fbonds_f32(f_atoms, g.nA, ld_fa, afd, g.b2a, fbond, ld_fbb, bfd, g.nB, fb, ld, st));
      g.f_bonds = fb; g.ld_fb = ld;
    }
    return g;
  };

  // ---- weights
  FILE* wf = fopen(argv[3], "rb");
  if (!wf) { perror("open weights"); return 1; }
  int32_t hd[7]; if (fread(hd, 4, 7, wf) != 7) return 1;
  const int H = hd[0], depth = hd[1], ddepth = hd[2], n_ffn = hd[3], F = hd[4], task_num = hd[5], head = hd[6];
  rr_model m; memset(&m, 0, sizeof(m));
  m.H = H; m.depth = depth; m.diff_depth = ddepth; m.n_ffn = n_ffn; m.head = head; m.atom_fdim = afd; m.bond_fdim = afd + bfd;
  std::vector<std::pair<int, int>> shapes;                    // (out, in) in file order
  auto lin = [&](rr_linear_w& L, int out, int in, bool present) {
    if (!present) return;
    std::vector<float> w(static_cast<size_t>(out) * in), b(out);
    if (fread(w.data(), 4, w.size(), wf) != w.size() || fread(b.data(), 4, b.size(), wf) != b.size()) { fprintf(stderr, "short weights file\n"); exit(1); }
    float* dw = dmalloc<float>(w.size()); float* db = dmalloc<float>(b.size());
    HIP_OK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    L.w = dw; L.b = db; L.out = out; L.in = in; L.ldw = in;
  };
  const int FB = afd + bfd;
  lin(m.enc_wi, H, FB, true); lin(m.enc_wh, H, H, depth > 1); lin(m.enc_wo, H, afd + H, true);
  lin(m.dif_wi, H, H, true); lin(m.dif_wh, H, H + FB, ddepth > 1); lin(m.dif_wo, H, 2 * H, ddepth > 0);
  for (int i = 0; i < n_ffn; ++i) lin(m.ffn[i], i == n_ffn - 1 ? task_num : H, i == 0 ? H + F : H, true);
  fclose(wf);

  // ---- the step
  rr_step s; memset(&s, 0, sizeof(s));
  s.p = graph("p", M, F32("p.f_atoms"), toc["p.f_atoms"].cols, F32("p.fbond"), toc["p.fbond"].cols, true);
  if (ddepth > 1) {                                             // sum_k f_bonds[a2b[a,k]] (models/mpn.py:202-209), input-only
    float* fs = dmalloc<float>(s.p.nA * s.p.ld_fb);
    HIP_OK(hipMemsetAsync(fs, 0, s.p.nA * s.p.ld_fb * 4, st));
    RR_OK_(rr_gather_sum_f32(s.p.f_bonds, s.p.nB, s.p.ld_fb, s.p.a2b, s.p.nA, s.p.K, FB, fs, s.p.ld_fb, st));
    s.p.fb_sum = fs; s.p.ld_fbs = s.p.ld_fb;
  }
  if (has_u) {
    rr_graph u = graph("u", toc["u.a_scope"].rows, F32("u.f_atoms"), toc["u.f_atoms"].cols, F32("u.fbond"), toc["u.fbond"].cols, true);
    if (drop_p == 0.f) {                                        // dropout inactive: encode every distinct reactant once
      s.mode = RR_STEP_DEDUP; s.r = u;
      s.amap = I32("amap"); s.amap_t = I32("amap_t"); s.amap_t_cols = static_cast<int>(toc["amap_t"].cols);
    } else {                                                    // train mode: share the deterministic prefix only
      const int64_t nAr = toc["r.a2b"].rows, ldfa = toc["u.f_atoms"].cols;
      float* fa = dmalloc<float>(nAr * ldfa);                   // the copies' atom features = the distinct reactants' rows
      RR_OK_(rr_gather_sum_f32(u.f_atoms, u.nA, ldfa, I32("amap"), nAr, 1, static_cast<int>(ldfa), fa, ldfa, st));
      s.mode = depth >= 2 ? RR_STEP_PREFIX : RR_STEP_PLAIN;
      s.u = u;
      bool need_fb = s.mode == RR_STEP_PLAIN;
      float* fbr = nullptr;
      if (need_fb) {
        const int64_t nBr = toc["r.b2a"].rows, ldb = toc["u.fbond"].cols;
        fbr = dmalloc<float>(nBr * ldb);
        RR_OK_(rr_gather_sum_f32(F32("u.fbond"), u.nB, ldb, I32("bmap"), nBr, 1, static_cast<int>(ldb), fbr, ldb, st));
      }
      s.r = graph("r", M, fa, ldfa, fbr, toc["u.fbond"].cols, need_fb);
      s.bmap = I32("bmap"); s.bmap_t = I32("bmap_t"); s.bmap_t_cols = static_cast<int>(toc["bmap_t"].cols);
    }
  } else {
    s.mode = RR_STEP_PLAIN;
    s.r = graph("r", M, F32("r.f_atoms"), toc["r.f_atoms"].cols, F32("r.fbond"), toc["r.fbond"].cols, true);
  }
  s.feat = F > 0 ? F32("add") : nullptr; s.F = F;
  s.drop_p = drop_p; s.seed = seed;
  float* out = dmalloc<float>(M * task_num);
  s.out = out;
  s.workspace_bytes = rr_reaction_workspace_bytes(&m, &s);
  if (s.workspace_bytes == 0) { fprintf(stderr, "rr_reaction_workspace_bytes rejected the step\n"); return 4; }
  s.workspace = dmalloc<uint8_t>(s.workspace_bytes);

  // plan flags: 0 = the three-exact-bf16-term GEMMs (no operand bit dropped, batch-independent scores) unless RR_CXX_PLAN_FLAGS
  // says otherwise (32 = RR_PLAN_F16X2_GEMM: two f16 terms per operand, half the matrix instructions, 22-bit operands scaled
  // per tensor - opt-in); forward and backward of a step get the same flags
  const char* fl_env = getenv("RR_CXX_PLAN_FLAGS");
  const int plan_flags = fl_env != nullptr ? atoi(fl_env) : 0;
  RR_OK_(rr_reaction_forward(&m, &s, plan_flags, st));
  // ListMLE over the step's queries (scores = first output column)
  std::vector<int32_t> seg(Q + 1, 0);
  const int32_t* scope = reinterpret_cast<const int32_t*>(blob + toc["scope"].off);
  int max_len = 0;
  for (int64_t q = 0; q < Q; ++q) { seg[q + 1] = seg[q] + scope[q]; if (scope[q] > max_len) max_len = scope[q]; }
  int32_t* dseg = dmalloc<int32_t>(Q + 1);
  HIP_OK(hipMemcpyAsync(dseg, seg.data(), (Q + 1) * 4, hipMemcpyHostToDevice, st));
  float *dloss = dmalloc<float>(1), *dpart = dmalloc<float>(Q), *dout = dmalloc<float>(M * task_num);
  unsigned int* dticket = dmalloc<unsigned int>(1);      // the step kernel's ticket word: zeroed once, the kernel re-arms it
  HIP_OK(hipMemsetAsync(dticket, 0, 4, st));
  HIP_OK(hipMemsetAsync(dout, 0, M * task_num * 4, st));
  // loss AND d loss / d score (the loss is the root of the graph: upstream gradient one) in ONE launch - the bits of
  // rr_listmle_fwd_f32 + rr_listmle_bwd_f32 with *gloss = 1 (ABI revision 8)
  RR_OK_(rr_listmle_step_f32(out, task_num, F32("targets"), dseg, static_cast<int>(Q), max_len, dloss, dpart, dticket, dout, task_num, st));
  rr_grads G; memset(&G, 0, sizeof(G));
  std::vector<std::pair<float*, size_t>> gbuf;
  const rr_linear_w* Ls[RR_G_FFN0 + RR_MAX_FFN] = {&m.enc_wi, &m.enc_wh, &m.enc_wo, &m.dif_wi, &m.dif_wh, &m.dif_wo};
  for (int i = 0; i < n_ffn; ++i) Ls[RR_G_FFN0 + i] = &m.ffn[i];
  for (int i = 0; i < RR_G_FFN0 + n_ffn; ++i) {
    if (!Ls[i]->w) { gbuf.push_back({nullptr, 0}); gbuf.push_back({nullptr, 0}); continue; }
    const size_t nw = static_cast<size_t>(Ls[i]->out) * Ls[i]->in;
    G.w[i] = dmalloc<float>(nw); G.b[i] = dmalloc<float>(Ls[i]->out);
    gbuf.push_back({G.w[i], nw}); gbuf.push_back({G.b[i], static_cast<size_t>(Ls[i]->out)});
  }
  RR_OK_(rr_reaction_backward(&m, &s, dout, &G, plan_flags, st));
  // The data-parallel exchange of a multi-GPU job (one process per GPU, whole queries per rank): ONE sum all-reduce per
  // gradient buffer over an RCCL communicator, scaled to the mean over equal shards.  This example is one process, so its
  // communicator has one rank and the exchange is the identity - the call sequence is what a rank of an N-GPU job runs
  // (rank 0 makes the id, every rank gets it over any host channel).  RR_CXX_DP=0 skips it (no RCCL on the machine).
  const char* dp_env = getenv("RR_CXX_DP");
  if (dp_env == nullptr || atoi(dp_env) != 0) {
    char id[RR_COMM_ID_BYTES];
    rr_comm_t comm = nullptr;
    const int n_ranks = 1, rank = 0;
    const int have = rr_comm_unique_id(id);
    if (have == RR_ERR_UNSUPPORTED) {                     // no RCCL on this machine: forward and backward stand as they are
      fprintf(stderr, "train_step: no RCCL found (rr_comm_backend() = %d), skipping the gradient exchange\n", rr_comm_backend());
    } else {
      RR_OK_(have);
      RR_OK_(rr_comm_init_rank(&comm, n_ranks, id, rank));
      for (auto& gb : gbuf)
        if (gb.first) RR_OK_(rr_allreduce_f32(gb.first, static_cast<int64_t>(gb.second), 1.0f / n_ranks, comm, st));
      HIP_OK(hipStreamSynchronize(st));
      RR_OK_(rr_comm_destroy(comm));
    }
  }
  HIP_OK(hipStreamSynchronize(st));

  float loss = 0.f;
  HIP_OK(hipMemcpy(&loss, dloss, 4, hipMemcpyDeviceToHost));
  std::vector<float> ho(M * task_num);
  HIP_OK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
  double so = 0; for (float v : ho) so += v;
  printf("{\"loss\": %.9g, \"out_sum\": %.17g, \"M\": %lld, \"mode\": %d, \"workspace_mb\": %.1f, \"grad_sums\": [", loss, so,
         static_cast<long long>(M), s.mode, s.workspace_bytes / 1e6);
  bool first = true;
  for (auto& gb : gbuf) {
    if (!gb.first) continue;
    std::vector<float> h(gb.second);
    HIP_OK(hipMemcpy(h.data(), gb.first, gb.second * 4, hipMemcpyDeviceToHost));
    double sg = 0; for (float v : h) sg += v;
    printf("%s%.17g", first ? "" : ", ", sg);
    first = false;
  }
  printf("]}\n");
  return 0;
}
