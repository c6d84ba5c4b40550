// Host-side graph packer: BatchMolGraph.__init__ (reference features/featurization.py:246-288)
// as one native pass over concatenated per-molecule arrays, plus the transposed index
// tables the backward kernels consume.  Pure C++ (no device code); all pointers are host.
//
// The reference builds these arrays with Python list appends and torch.LongTensor(list)
// (~0.35 ms per molecule); here it is a memcpy-speed loop.
#include <cstring>

#include "rr_common.h"

namespace {

int compute_k(const int64_t* a2b_off, int64_t total_atoms) {
  int64_t kmax = 1;                                     // max(1, ...)  featurization.py:281
  for (int64_t a = 0; a < total_atoms; ++a) {
    const int64_t d = a2b_off[a + 1] - a2b_off[a];
    if (d > kmax) kmax = d;
  }
  return static_cast<int>(kmax);
}

}  // namespace

extern "C" {

int rr_pack_sizes(const int32_t* mol_atoms, const int32_t* mol_bonds, int64_t M, const int64_t* a2b_off,
                  int K_override, int64_t* nA, int64_t* nB, int32_t* K) {
  RR_CHECK_ARG(M >= 0 && (M == 0 || (mol_atoms && mol_bonds)) && a2b_off && nA && nB && K && K_override >= 0);
  int64_t ta = 0, tb = 0;
  for (int64_t i = 0; i < M; ++i) {
    RR_CHECK_ARG(mol_atoms[i] >= 0 && mol_bonds[i] >= 0);
    ta += mol_atoms[i];
    tb += mol_bonds[i];
  }
  const int kmax = compute_k(a2b_off, ta);
  if (K_override != 0 && K_override < kmax) return RR_ERR_ARG;
  *nA = ta + 1;                                         // +1: padding row 0, featurization.py:255-256
  *nB = tb + 1;
  *K = K_override != 0 ? K_override : kmax;
  RR_CHECK_ARG(*nA < INT32_MAX && *nB < INT32_MAX);
  return RR_OK;
}

int rr_derive_tables(const int32_t* a2b, const int32_t* b2a, const int32_t* b2revb, int64_t nA, int64_t nB, int K,
                     const int32_t* a_scope, int64_t M, int32_t* a2a, int32_t* a2b_rev_t, int32_t* b2t,
                     int32_t* a2a_t, float* npad, int32_t* atom2mol) {
  RR_CHECK_ARG(a2b && b2a && b2revb && nA >= 1 && nB >= 1 && K >= 1 && M >= 0 && (M == 0 || a_scope));
  for (int64_t a = 0; a < nA; ++a) {
    int pads = 0;
    for (int k = 0; k < K; ++k) {
      const int32_t b = a2b[a * K + k];
      RR_CHECK_ARG(b >= 0 && b < nB);
      const bool pad = (b == 0);                        // bond 0 is the padding bond
      if (pad) ++pads;
      if (a2a) a2a[a * K + k] = b2a[b];                 // get_a2a, featurization.py:326-327
      if (a2b_rev_t) a2b_rev_t[a * K + k] = pad ? -1 : b2revb[b];
      if (a2a_t) a2a_t[a * K + k] = pad ? -1 : b2a[b];
    }
    if (npad) npad[a] = static_cast<float>(pads);       // row 0 has K pads
  }
  // row 0: bond 0 is the only bond whose source is atom 0 (b2a[0] = 0)
  if (a2b_rev_t) a2b_rev_t[0] = 0;
  if (b2t) {
    b2t[0] = -1;                                        // row 0 is rebuilt by the pad-row reduction
    for (int64_t b = 1; b < nB; ++b) {
      const int32_t r = b2revb[b];
      RR_CHECK_ARG(r >= 0 && r < nB);
      b2t[b] = b2a[r];                                  // the atom bond b points to
    }
  }
  if (atom2mol) {
    for (int64_t a = 0; a < nA; ++a) atom2mol[a] = -1;
    for (int64_t m = 0; m < M; ++m) {
      const int32_t start = a_scope[2 * m], size = a_scope[2 * m + 1];
      RR_CHECK_ARG(start >= 0 && size >= 0 && static_cast<int64_t>(start) + size <= nA);
      for (int32_t i = 0; i < size; ++i) atom2mol[start + i] = static_cast<int32_t>(m);
    }
  }
  return RR_OK;
}

int rr_derive_bond_tables(const int32_t* a2b_rev_t, const int32_t* b2t, const int32_t* b2revb, const int32_t* b2a,
                          const float* npad, int64_t nA, int64_t nB, int K, int Kb, int32_t* b2b_t, float* npad_b) {
  RR_CHECK_ARG(a2b_rev_t && b2t && b2revb && b2a && npad && nA >= 1 && nB >= 1 && K >= 1 && Kb >= 1 && Kb >= K - 1);
  RR_CHECK_ARG(b2b_t && npad_b);
  // row 0 (padding bond): rebuilt by the weighted column sum, no gathered sources
  for (int j = 0; j < Kb; ++j) b2b_t[j] = -1;
  npad_b[0] = npad[0] - 1.0f;                           // d_amsg[0] = d_min[0] is read K times, minus d_min[rev(0) = 0]
  for (int64_t b = 1; b < nB; ++b) {
    const int32_t t = b2t[b], r = b2revb[b], src = b2a[b];
    RR_CHECK_ARG(t >= 0 && t < nA && src >= 0 && src < nA);
    int n = 0;
    for (int k = 0; k < K; ++k) {
      const int32_t o = a2b_rev_t[static_cast<int64_t>(t) * K + k];   // bonds leaving the atom b points to
      if (o < 0 || o == r) continue;
      if (n >= Kb) return RR_ERR_ARG;
      b2b_t[b * Kb + n++] = o;
    }
    for (; n < Kb; ++n) b2b_t[b * Kb + n] = -1;
    npad_b[b] = npad[src];
  }
  return RR_OK;
}

int rr_pack_graphs(const int32_t* mol_atoms, const int32_t* mol_bonds, int64_t M, const float* f_atoms_cat,
                   int atom_fdim, const float* f_bonds_cat, int bond_fdim, const int32_t* b2a_local,
                   const int32_t* b2revb_local, const int64_t* a2b_off, const int32_t* a2b_local, int K,
                   float* f_atoms, int64_t ld_fa, float* f_bonds, int64_t ld_fb, int32_t* a2b, int32_t* b2a,
                   int32_t* b2revb, int32_t* a2a, int32_t* a_scope, int32_t* a2b_rev_t, int32_t* b2t, int32_t* a2a_t,
                   float* npad, int32_t* atom2mol) {
  RR_CHECK_ARG(M >= 0 && atom_fdim >= 1 && bond_fdim >= 1 && K >= 1 && ld_fa >= atom_fdim && ld_fb >= bond_fdim);
  RR_CHECK_ARG(f_atoms && f_bonds && a2b && b2a && b2revb && a_scope && a2b_off);
  RR_CHECK_ARG(M == 0 || (mol_atoms && mol_bonds && f_atoms_cat));
  // padding row 0 (featurization.py:260-264): zero features, a2b[0] = [0]*K, b2a[0] = b2revb[0] = 0
  std::memset(f_atoms, 0, sizeof(float) * ld_fa);
  std::memset(f_bonds, 0, sizeof(float) * ld_fb);
  for (int k = 0; k < K; ++k) a2b[k] = 0;
  b2a[0] = 0;
  b2revb[0] = 0;
  int64_t na = 1, nb = 1, ca = 0, cb = 0;               // batch offsets / concatenated-input offsets
  for (int64_t m = 0; m < M; ++m) {
    const int32_t ma = mol_atoms[m], mb = mol_bonds[m];
    for (int32_t i = 0; i < ma; ++i) {
      float* dst = f_atoms + (na + i) * ld_fa;
      std::memcpy(dst, f_atoms_cat + (ca + i) * atom_fdim, sizeof(float) * atom_fdim);
      if (ld_fa > atom_fdim) std::memset(dst + atom_fdim, 0, sizeof(float) * (ld_fa - atom_fdim));
      const int64_t lo = a2b_off[ca + i], hi = a2b_off[ca + i + 1];
      if (hi - lo > K) return RR_ERR_ARG;
      int32_t* row = a2b + (na + i) * K;
      int k = 0;
      for (int64_t j = lo; j < hi; ++j, ++k) {
        if (a2b_local[j] < 0 || a2b_local[j] >= mb) return RR_ERR_ARG;
        row[k] = static_cast<int32_t>(nb + a2b_local[j]);   // featurization.py:269
      }
      for (; k < K; ++k) row[k] = 0;                        // right-pad with 0, featurization.py:286
    }
    for (int32_t j = 0; j < mb; ++j) {
      float* dst = f_bonds + (nb + j) * ld_fb;
      std::memcpy(dst, f_bonds_cat + (cb + j) * bond_fdim, sizeof(float) * bond_fdim);
      if (ld_fb > bond_fdim) std::memset(dst + bond_fdim, 0, sizeof(float) * (ld_fb - bond_fdim));
      if (b2a_local[cb + j] < 0 || b2a_local[cb + j] >= ma) return RR_ERR_ARG;
      if (b2revb_local[cb + j] < 0 || b2revb_local[cb + j] >= mb) return RR_ERR_ARG;
      b2a[nb + j] = static_cast<int32_t>(na + b2a_local[cb + j]);        // featurization.py:273
      b2revb[nb + j] = static_cast<int32_t>(nb + b2revb_local[cb + j]);  // featurization.py:274
    }
    a_scope[2 * m] = static_cast<int32_t>(na);              // featurization.py:276
    a_scope[2 * m + 1] = ma;
    na += ma;
    nb += mb;
    ca += ma;
    cb += mb;
  }
  return rr_derive_tables(a2b, b2a, b2revb, na, nb, K, a_scope, M, a2a, a2b_rev_t, b2t, a2a_t, npad, atom2mol);
}

}  // extern "C"
