// ff_eloc_mfma.h -- local-energy sensitivities on the fp64 matrix cores (included by ff_cnf_fwd.hip).
//
// Same system as MODE 2 of ff_ode_fwd_kernel (src/VMC.py:46-49 / src/utils.py:40-65 by forward sensitivities):
//     z' = v(z)            J' = A J   (A = dv/dz, J = dz/dx)         kbar' = A kbar + sum_i D2v[u_i, u_i]
//     Delta' = -div v      (grad Delta)' = -J^T g  (g = grad_z div v)  (lap Delta)' = -(sum_i D2div[u_i, u_i] + g . kbar)
// (u_i: columns of J).  Per right-hand side the dense work is two M x M x M products, J' = A J and S = J J^T -- the
// second-order sources need nothing else of J: for a pair term eta(|rho|) rho with W = S_aa + S_bb - S_ab - S_ba,
//     sum_i D2[delta_i, delta_i] = (2 eta'/r) W rho + [eta'' q + eta' (tr W - q)/r] rho,   q = rho^T W rho / r^2,
//     sum_i D2div[...]           = phi'' q + phi' (tr W - q)/r                            (phi = c (eta' r + D eta))
// -- and they run on v_mfma_f64_4x4x4_4b_f64: FOUR walkers per wave, one per matrix-instruction block.
//
// Lanes: l = 16 r + 4 w + c  (w: walker slot = the instruction's block, r, c in 0..3).  M = N*D <= 12 coordinates are
// padded to MP = 4 MB; the lane holds element (r, c) of every 4 x 4 block of J: Jb[I][K] = J[4I + r][4K + c]
// (= the instruction's B and C/D layout), MB^2 doubles.
//   J' block (I, Kc)  = sum_K  mfma(A-operand of A(I,K), Jb[K][Kc]);  A is symmetric, so that operand -- A[4I+c][4K+r] --
//                       is A[4K+r][4I+c]: element (r, c) of block (K, I), read straight from the assembled matrix in LDS;
//   S  block (I, Kc)  = sum_K  mfma(Jt[I][K], Jt[Kc][K]),  Jt = the blocks transposed in place (lane (r,c) <-> (c,r)) -- by the
//                       matrix cores themselves: a lane's element fed as the A operand IS the transposed block, one product with the
//                       identity returns it (round 6; the LDS crossbar took two ds_bpermute per double); only I <= Kc is computed.
// Everything else is per-radius or per-coordinate work, without atomics:
//   R1  radius lanes -- the P PAIR radii of a walker are dealt over ITS sixteen lanes: heads from the radial table, one record per
//       radius (rho, eta, eta'/r, c phi'/r, the first-order part of D_v[kbar] + the second-order source) in LDS; the radius' shares
//       of div v and of the Laplacian source stay on the lane (Delta and lap Delta are integrated as per-lane partial sums, joined
//       over the walker's quads by a product with ones);
//       the ONE-BODY radius of a particle is evaluated by that particle's own row lanes and applied from registers (no record);
//   rows  lane (r, c) owns coordinate p = 4r + c (< M) of particle a = p / D: it walks the N - 1 pair records of its particle
//       (either sign), sums v_p, (A kbar)_p, (grad div)_p and the diagonal block of A, and stores row p of A -- every element
//       of A is written by exactly one lane;
//   R2  radius lanes: W from S, the second-order sources written back into the record; the row lanes gather them.
//   The six stages of a Dormand-Prince step are six instantiations of the evaluation (compile-time stage: DESIGN.md 3s).
//   S takes the place of A in LDS once the last operand of J' = A J has been read (one wave per workgroup: program order).
//   The component 4c + r of grad Delta sits on lane (r, c) (it comes out of the transposed blocks by a quad reduction).
// 14 KB of LDS and 236 registers: two waves per SIMD (the vector pipe issues at half rate for a single wave).
#pragma once

// Dormand-Prince vector of a lane whose components 1 .. NB (the J blocks) live in lane-private LDS columns and whose other
// NS components (slot 0 and NB + 1 ...) live in registers
// Weight of Delta and lap Delta in the error norm relative to the other sensitivity components.  The column and row kernels carry
// both as per-lane partial sums, each controlled against its own magnitude -- a norm that is stricter by the square root of the
// number of partials (more where they cancel); here each is ONE component and the weight stands in for that.  It decides how many
// steps the walkers with a particle passing the origin take (the kink of mu(|x|) x), i.e. the length of the longest chain of the
// launch, against their E_loc error (65 536 walkers, config 2, sweep policy; tools/probes/sens_tol.py):
//   weight 1: max 2.9e-6, p99.9 1.6e-7, launch 0.79 ms | 4: 2.1e-6, 5.5e-8, 0.89 ms | 16: 2.1e-6, 1.3e-8, 1.05 ms
//   (column kernel: 2.1e-7, 1.7e-8, 1.03 ms; the bar is 1e-5)
// The weight is ff_ode.sum_weight (A.sum_w; the entry point substitutes the default, 4, for 0).
template <int NB, int NS>
struct ff_jsplit_vec {
  double* col;
  double r[NS];
  FF_D ff_jsplit_vec(double* base, int lane) : col(base + lane) {}
  FF_D double get(int v) const { return (v >= 1 && v <= NB) ? col[(v - 1) * FF_WAVE] : r[v == 0 ? 0 : v - NB]; }
  FF_D void set(int v, double x) { if (v >= 1 && v <= NB) col[(v - 1) * FF_WAVE] = x; else r[v == 0 ? 0 : v - NB] = x; }
};

template <int N, int D, bool TAB, int WPS = 1>
__global__ void __launch_bounds__(FF_WAVE, WPS)
ff_eloc_mfma_kernel(ff_fwd_args A) {
  constexpr int M = N * D, MB = (M + 3) / 4, MP = 4 * MB, G = 4;
  static_assert(MB >= 1 && MB <= 3, "at most 12 coordinates per walker");
  constexpr int P = N * (N - 1) / 2, R = P + N;
  // components of a lane: 0: z_p on the coordinate lanes p < M, Delta on lane p = 12, lap Delta on lane p = 13 (M <= 12: never
  // coordinate lanes); 1 .. NB: the J blocks; kbar_p; (grad Delta)_p
  constexpr int NH = 4, NB = MB * MB, NV = NB + 3;
  constexpr int IK = NB + 1, IDD = NB + 2, PDL = 12, PLP = 13;

  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  constexpr int AS = MP + 1;     // row stride of the per-walker matrix (odd: the column reads of the operands spread over the banks)
  __shared__ double s_A[G][MP * AS];       // A = dv/dz, then S = J J^T
  __shared__ double s_z[G][MP], s_kb[G][MP];   // per coordinate: z, kbar (published); grad div (row lanes) takes the place of z
  // step-size controller of a walker: kept in LDS, loaded by the walker's lanes where a decision is taken (every lane of the
  // walker computes the same update and stores the same values) -- forty registers less in the right-hand side
  struct ctl_t {
    ff_stepper S; double hmax_acc, h0v, d1v, hwarm, sens_w;
    FF_D void get(const ctl_t& o) {      // member by member (a struct copy is a memcpy through scratch)
      S.t = o.S.t; S.tb = o.S.tb; S.dir = o.S.dir; S.interval = o.S.interval; S.habs = o.S.habs; S.h = o.S.h; S.tnew = o.S.tnew;
      S.hprev = o.S.hprev; S.eprev = o.S.eprev; S.nacc = o.S.nacc; S.nrej = o.S.nrej; S.natt = o.S.natt; S.rejected = o.S.rejected;
      S.fail = o.S.fail; S.done = o.S.done;
      hmax_acc = o.hmax_acc; h0v = o.h0v; d1v = o.d1v; hwarm = o.hwarm; sens_w = o.sens_w;
    }
  };
  __shared__ ctl_t s_ctl[G];
  // record of a radius: [0,D) rho  [D] eta  [D+1] eta'/r  [D+2] c phi'/r  [D+3, 2D+3) this radius' term of A kbar + the second-order source
  constexpr int RW = (2 * D + 3) | 1;     // odd: the row lanes of a wave read up to 4 N records at a time, one bank group each
  constexpr int QF0 = D, QF1 = D + 1, QGQ = D + 2, QPW = D + 3;
  __shared__ double s_rec[(G * P + 1) * RW];   // one record per PAIR radius + one that stays zero
#ifndef FF_MFMA_Y_COLS
#define FF_MFMA_Y_COLS 0      // 1: y's J blocks in LDS columns as in rounds 3-6 (A/B)
#endif
  __shared__ double s_cv[(FF_MFMA_Y_COLS ? 2 : 1) * NB][FF_WAVE];   // the J blocks of the error accumulator of the Dormand-Prince step (and, FF_MFMA_Y_COLS, of y), lane-private columns
  __shared__ int s_any;
  __shared__ int s_st[4];
  __shared__ long long s_next;
  // where the lanes without a coordinate (p >= M) put what the row lanes store: no exec masks around the thirteen stores of that phase
  // (round 6: 236 -> 207 instructions, -0.8 % of the pass)
  __shared__ double s_dump[MP + 2];

  const int lane = threadIdx.x;
  const int r = lane >> 4, w = (lane >> 2) & 3, c = lane & 3;
  const int p = 4 * r + c;                 // the coordinate this lane integrates (if < M)
  const bool owner = p < M;
  const int tl = 16 * c + 4 * w + r;       // the lane holding the transposed block element
  const double* __restrict__ rtab = A.net.radial_table;
  if constexpr (TAB) {
    if (rtab[3] != 0.0) {   // table unusable for these weights: leave the call to the direct kernel
      if (lane == 0 && blockIdx.x == 0) *A.evt = A.evt_id;
      return;
    }
  } else {
    if (A.evt && *A.evt != A.evt_id) return;   // fallback launch that is not needed
    ff_fill_exp2_table(s_e2, lane);
    ff_load_weights(s_w, A.net, lane);
  }
  bool off_table = false;
  if (lane < 4) s_st[lane] = 0;
  for (int e = lane; e < G * MP * AS; e += FF_WAVE) (&s_A[0][0])[e] = 0.0;   // padding stays zero
  for (int e = lane; e < G * MP; e += FF_WAVE) { (&s_z[0][0])[e] = 0.0; (&s_kb[0][0])[e] = 0.0; }
  for (int e = lane; e < (G * P + 1) * RW; e += FF_WAVE) s_rec[e] = 0.0;
  FF_WG1_SYNC();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  const double rtol = A.rtol, atol = A.atol;
  double (*s_gd)[MP] = s_z;
  constexpr double NT = (double)M * M + 4.0 * M + 2.0;   // z, J, kbar, grad Delta, Delta, lap Delta
  const int64_t ngroups = (A.B + G - 1) / G;
  // Pair radii this lane evaluates for its own walker (slot qk: pair i16 + 16 qk): a | b << 4 | record offset << 8.  The ONE-BODY
  // radius of a particle is evaluated by that particle's own row lanes (both of them, redundantly) and applied from registers: no
  // record, no second radius slot at six particles -- per wave-evaluation seven LDS stores and six loads less on a kernel that is
  // bound by its LDS operations.
  constexpr int NQ = (P + 15) / 16;
  const int i16 = 4 * r + c;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) {
    const int pr = i16 + 16 * qk;
    int a = 0, bb = 1;
    if (pr < P) {
      int q = pr;
      while (q >= N - 1 - a) { q -= N - 1 - a; a++; }
      bb = a + 1 + q;
    }
    rq_id[qk] = pr < P ? (a | (bb << 4) | (((w * P + pr) * RW) << 8)) : -1;
  }
  // row lanes: partner slot j = particle k(j) = j + (j >= own particle), j < N - 1.  Packed: (k < own) | record offset << 1 |
  // byte offset of block k in the lane's row of A << 13.  A lane without a coordinate reads the all-zero record behind the last one.
  const int ra = owner ? p / D : 0, rc = owner ? p % D : 0;
  constexpr int NPART = N - 1 > 0 ? N - 1 : 1;
  int prec[NPART];
#pragma unroll
  for (int j = 0; j < NPART; j++) {
    const int k = j + (j >= ra ? 1 : 0);
    const bool ok = owner && j < N - 1;
    const int lo = k < ra ? k : ra, hi = k < ra ? ra : k;
    const int pr = lo * (2 * N - lo - 1) / 2 + (hi - lo - 1);
    prec[j] = ((ok ? (w * P + pr) : G * P) * RW) << 1 | ((ok && k < ra) ? 1 : 0) | ((ok ? k * D * 8 : 0) << 13);
  }

  // sum over the 16 lanes of a walker (identical on all of them): quad by DPP, the four quads by the matrix cores -- with ones as the A
  // operand a 4 x 4 x 4 product returns the column sums of B, sum_k B_blk[k][j], on every lane (i, j) of the block: the sum over the
  // walker's four quads r = k, in the fixed order k = 0 .. 3 (the LDS crossbar took two exchanges = four ds_bpermute for it)
  auto quads_sum = [&](double part) -> double { return ff_mfma4(1.0, part, 0.0); };
  auto walker_sum = [&](double part) -> double {
    part += ff_swap1(part);
    part += ff_swap2(part);
    return quads_sum(part);
  };

#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#ifdef FF_STAMPS_TRACE
  unsigned long long stage_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stage_cnt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#endif
  for (int64_t grp = blockIdx.x;; grp += gridDim.x) {
    if (A.queue) {   // persistent grid: next group from the launch's work counter (the order is by schedule key: cost class + 4 x planned steps, costliest first)
      FF_WG1_SYNC();
      if (lane == 0) s_next = (long long)atomicAdd(A.queue + (TAB ? 0 : 1), 1ULL);
      FF_WG1_SYNC();
      grp = s_next;
    }
    if (grp >= ngroups) break;
    // which walker this lane serves: derived from the (scalar) group index wherever it is needed -- here and again behind the
    // evaluation loop -- so that no per-lane copy of it is live through the loop (at 256 registers such values are spilled around it:
    // with the fused finish the scratch traffic of ~50 of them showed as 120 MB of HBM writes per launch)
    auto walker_of = [&](int wq, int64_t& bw, bool& vw) {
      const int64_t bq = grp * G + wq;
      const bool inb = bq < A.B;
      bw = ff_opt_load(A.order, inb, bq, A.y_in, (int32_t)bq);
      // routing by cost class (launch_mfma): with heavy_mode = 2 the walkers of class >= heavy_class belong to another launch
      vw = inb && !(A.heavy_mode == 2 && ff_opt_load(A.wclass, inb, bw, A.y_in, (int32_t)0) >= A.heavy_class);
    };
    int lgp = lane;
    FF_OPAQUE(lgp);
    const int r = lgp >> 4, w = (lgp >> 2) & 3, c = lgp & 3, p = 4 * r + c;
    const bool owner = p < M;
    int64_t b;
    bool valid;
    walker_of(w, b, valid);
    // Dormand-Prince storage as in ff_ode_fwd_kernel: y, c0..c2 (k0..k2, then the inputs of stages 4, 5 and y_new), c3 (error)
    double c0[NV], c1[NV], c2[NV];
    // y in registers (round 6, late: with compile-time stages the kernel runs at 208 registers, and y's J blocks -- read from their LDS
    // columns in every evaluation of stages 1-3, written at every accepted step -- fit: 238); the error accumulator c3, written at
    // stage 3 and read at stage 4, keeps its J blocks in lane-private LDS columns (in registers as well: 256 + 68 B of scratch)
    ff_jsplit_vec<FF_MFMA_Y_COLS ? NB : 0, FF_MFMA_Y_COLS ? 3 : NV> y(&s_cv[0][0], lane);
    ff_jsplit_vec<NB, 3> c3(&s_cv[FF_MFMA_Y_COLS ? NB : 0][0], lane);
#pragma unroll
    for (int v = 0; v < NV; v++) { y.set(v, 0.0); c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3.set(v, 0.0); }
    {
      const double y0 = ff_opt_load(A.y_in, valid && owner, b * M + p, A.y_in, 0.25 * (p + 1) + 0.125 * ((p * 7) % 5));   // idle walkers: finite, distinct
      if (owner) y.set(0, y0);
    }
#pragma unroll
    for (int I = 0; I < MB; I++) y.set(1 + I * MB + I, (r == c && 4 * I + r < M) ? 1.0 : 0.0);   // J = identity
    {
      ctl_t C;
      C.S.begin(A.ta, A.tb, valid);
      // walkers of a low cost class: looser tolerance for the sensitivity components, larger first step (ff_ode.walker_class)
      const bool loose = ff_opt_load(A.wclass, valid, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
      C.hwarm = ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale));
      if (!(C.hwarm > 0.0)) C.hwarm = 0.0;
      // tolerance of the sensitivity components relative to the coordinates' (ff_ode.sens_tol): weight in the error norm
      C.sens_w = loose ? A.sens_w : 1.0;
      C.hmax_acc = 0.0; C.h0v = 0.0; C.d1v = 0.0;
      FF_WG1_SYNC();
      s_ctl[w].get(C);
      FF_WG1_SYNC();
    }
    int s = -2, nev = 0;
    double hs_keep = 0.0;      // = s_ctl[w].S.h of the attempt under way

    // One evaluation of the right-hand side and what the Dormand-Prince step does with it.  The six stages of a step are SIX
    // instantiations of this body (SG = 1 .. 6: the stage is a compile-time constant; straight-line code between them), the rare
    // evaluations in front of a step -- the first one of a walker group (-2), the probe of Hairer's rule (-1), k0 again after a rejected
    // step (0) -- share a seventh (SG = FF_STAGE_DYN: the stage is the run-time value s).  With ONE body and a run-time stage (rounds
    // 3-6) every evaluation ended in a merge of nine branches and a loop back-edge, and the compiler moved the whole Dormand-Prince
    // state -- 42 doubles per lane -- through two sets of registers on the way: 84 v_mov_b64 per evaluation, a tenth of the kernel's
    // vector instructions; a stage known at compile time also knows which of c0 / c1 / c2 enter its input.
    // Returns true when every walker of the wave has finished.
    auto evaluate = [&](auto stage_tag) -> bool {
      constexpr int SG = decltype(stage_tag)::value;
      auto at = [&](int k) -> bool { if constexpr (SG == FF_STAGE_DYN) return s == k; else return SG == k; };
      auto upto3 = [&]() -> bool { if constexpr (SG == FF_STAGE_DYN) return true; else return SG <= 3; };
      // the lane indices are laundered once per evaluation: otherwise every LDS address below becomes a loop-invariant register
      int ln = FF_LANE_SELF();
      FF_OPAQUE(ln);
      const int r = ln >> 4, w = (ln >> 2) & 3, c = ln & 3, p = 4 * r + c, tl = 16 * c + 4 * w + r;
      const bool owner = p < M;
      const int ra = owner ? p / D : 0, rc = owner ? p % D : 0;
      // ------------------------------------------------------------------ stage input
      // The walker's step size comes from a register copy of s_ctl[w].S.h (no LDS round trip in front of an evaluation), and
      // g_k = hs a_k + b_k with the stage's tableau entries a_k and b_k in {0, 1}: constants in the per-stage instantiations, scalar
      // selects on the (wave-uniform) stage index in the run-time one
#ifdef FF_HS_FROM_LDS
      const double hs = s_ctl[w].S.h;
#else
      const double hs = hs_keep;
#endif
      const double a0 = at(1) ? FF_A10 : (at(2) ? FF_A20 : (at(3) ? FF_A30 : 0.0));
      const double a1 = at(2) ? FF_A21 : (at(3) ? FF_A31 : 0.0);
      const double a2 = at(3) ? FF_A32 : 0.0;
      double g0 = fma(hs, a0, at(4) ? 1.0 : 0.0);
      const double g1 = fma(hs, a1, at(5) ? 1.0 : 0.0), g2 = fma(hs, a2, at(6) ? 1.0 : 0.0);
      if (at(-1)) g0 = s_ctl[w].h0v * s_ctl[w].S.dir;      // (cold start only: the probe evaluation of Hairer's rule)
      // (with a compile-time stage the vanishing terms are not formed: y + g0 c0 at stage 1, c0 alone at stage 4, ...)
      auto form = [&](int v) -> double {
        if constexpr (SG == FF_STAGE_DYN) return fma(g0, c0[v], y.get(v));      // stages -2, 0: g0 = 0; the probe: h0 k0
        else if constexpr (SG == 1) return fma(g0, c0[v], y.get(v));
        else if constexpr (SG == 2) return fma(g1, c1[v], fma(g0, c0[v], y.get(v)));
        else return fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], y.get(v))));
      };
      // Stages 4-6 take their input from c0 / c1 / c2 alone (with a compile-time stage: from the one array that holds it).
      const bool use_y = upto3();
      auto form_noy = [&](int v) -> double {
        if constexpr (SG == 4) return c0[v];
        else if constexpr (SG == 5) return c1[v];
        else return c2[v];
      };
      double out[NV];
      const double zin = use_y ? form(0) : form_noy(0), kin = use_y ? form(IK) : form_noy(IK);
      FF_STAMP(0);
      FF_SCHED_FENCE();
      // ------------------------------------------------------------------ publish z, kbar
      FF_WG1_SYNC();
      if (owner) { s_z[w][p] = zin; s_kb[w][p] = kin; }
      FF_WG1_SYNC();
      // ------------------------------------------------------------------ radius lanes (pair radii of the lane's own walker) and
      // row lanes (the one-body radius of the lane's own particle), first half: the radii, their table rows requested -- the fetch
      // runs under the S product
      double rq_rho[NQ][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
      bool rq_ok[NQ];
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        int id = rq_id[qk];
        FF_OPAQUE(id);
        const bool act = id >= 0;
        const int a = act ? (id & 15) : 0, bb = act ? ((id >> 4) & 15) : 1;
        double r2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < D; cc++) {
          rq_rho[qk][cc] = s_z[w][a * D + cc] - s_z[w][bb * D + cc];
          r2 = fma(rq_rho[qk][cc], rq_rho[qk][cc], r2);
        }
        if (!act) r2 = 1.0;
        ff_sqrt_rcp(r2, rq_r[qk], rq_ri[qk]);
        rq_dr[qk] = 0.0;
        rq_ok[qk] = true;
        if constexpr (TAB) {
          rq_ok[qk] = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, 0, rq_r[qk], rq_T[qk], rq_dr[qk]);
          if (act && !rq_ok[qk]) off_table = true;
        }
      }
      const bool ob_act = owner && has_mu;
      double ob_rho[D], ob_r, ob_ri, ob_T[TAB ? NH + 5 : 1], ob_dr = 0.0;
      bool ob_ok = true;
      {
        double r2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < D; cc++) {
          ob_rho[cc] = s_z[w][ra * D + cc];
          r2 = fma(ob_rho[cc], ob_rho[cc], r2);
        }
        if (!ob_act) r2 = 1.0;
        ff_sqrt_rcp(r2, ob_r, ob_ri);
        if constexpr (TAB) {
          ob_ok = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, 1, ob_r, ob_T, ob_dr);
          if (ob_act && !ob_ok) off_table = true;
        }
      }
      // ------------------------------------------------------------------ S = J J^T on the matrix cores
      // (stages 1-3: the stage's J blocks are formed ONCE -- y's part read from its LDS columns once -- and kept for J' = A J below:
      // eighteen registers the kernel has had to spare since its stages are compile-time constants, 251 -> 208)
      double Jk[NB];
      {
        // The blocks transposed in place, lane (r, c) <-> (c, r) -- by the matrix cores: a lane's element fed as the A operand IS the
        // transposed block (A_blk[i][k] is supplied by lane (k, i)), so one product with the identity returns it in the C/D layout,
        // exactly (one non-zero product per sum).  Nine matrix instructions on a pipe that is 10 % busy instead of eighteen ds_bpermute
        // on the LDS crossbar this kernel is bound by.
        double Jt[NB];
        const double idn = r == c ? 1.0 : 0.0;
        if (use_y) {
#pragma unroll
          for (int e = 0; e < NB; e++) { Jk[e] = form(1 + e); Jt[e] = ff_block_transpose(Jk[e], idn, tl); }
        } else {
#pragma unroll
          for (int e = 0; e < NB; e++) Jt[e] = ff_block_transpose(form_noy(1 + e), idn, tl);
        }
        double* Sw = s_A[w];
#pragma unroll
        for (int I = 0; I < MB; I++) {
#pragma unroll
          for (int Kc = I; Kc < MB; Kc++) {
            double acc = 0.0;
#pragma unroll
            for (int K = 0; K < MB; K++) acc = ff_mfma4(Jt[I * MB + K], Jt[Kc * MB + K], acc);
            Sw[(4 * I + r) * AS + 4 * Kc + c] = acc;
            // (the mirrored half of an off-diagonal block is read by nobody in the right-hand side: W needs S_aa, S_bb -- inside the
            // diagonal blocks, D divides 4 -- and S_ab with a < b, the upper triangle; the finish forms its own S, both halves)
            if (Kc != I && 4 % D != 0) Sw[(4 * Kc + c) * AS + 4 * I + r] = acc;
          }
        }
      }
      FF_WG1_SYNC();
      FF_STAMP(1);
      FF_SCHED_FENCE();
      // ------------------------------------------------------------------ radius lanes, second half: heads, contraction of the
      // second-order terms with S, one record per pair radius; the row lanes keep their one-body radius' terms in registers
      double dsum = 0.0, qsum = 0.0;
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        int id = rq_id[qk];
        FF_OPAQUE(id);
        const bool act = id >= 0;
        const int a = act ? (id & 15) : 0, bb = act ? ((id >> 4) & 15) : 1;
        const double* rho = rq_rho[qk];
        const double rr = rq_r[qk], ri = rq_ri[qk];
        double hd[NH];
        if constexpr (TAB) {
          if (rq_ok[qk]) ff_table_eval<NH>(rq_T[qk], rq_dr[qk], hd);
          else {
#pragma unroll
            for (int m = 0; m < NH; m++) hd[m] = 0.0;
          }
        } else {
          ff_heads<NH, true>(s_w[0], s_e2, He, rr, hd);
        }
        if (act) {
          const double f0 = hd[0], f1 = hd[1], f2 = hd[2], f3 = hd[3];
          const double Ac = 2.0 * fma(f2, rr, (1.0 + D) * f1), Bc = 2.0 * fma(f3, rr, (2.0 + D) * f2);
          const double f1ri = f1 * ri, gq = Ac * ri;
          // W = S_aa + S_bb - S_ab - S_ba;  w1 = W rho, q = rho^T W rho / r^2
          const double* Sg = s_A[w];
          double w1[D], dk[D], qq = 0.0, tr = 0.0, rdk = 0.0;
          // (W is symmetric -- S is, element for element: both halves of a block pair are one accumulator, and the two lanes of a
          // diagonal block form the same products in the same order -- so only c2i >= cc is read: 12 LDS reads per pair instead of 16)
          double Ws[D][D];
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
#pragma unroll
            for (int c2i = cc; c2i < D; c2i++) {
              double ww = Sg[(a * D + cc) * AS + a * D + c2i];
              ww += (Sg[(bb * D + cc) * AS + bb * D + c2i] - Sg[(a * D + cc) * AS + bb * D + c2i]) - Sg[(a * D + c2i) * AS + bb * D + cc];
              Ws[cc][c2i] = ww; Ws[c2i][cc] = ww;
            }
          }
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            double t = 0.0;
#pragma unroll
            for (int c2i = 0; c2i < D; c2i++) {
              const double ww = Ws[cc][c2i];
              t = fma(ww, rho[c2i], t);
              if (c2i == cc) tr += ww;
            }
            w1[cc] = t;
            qq = fma(rho[cc], t, qq);
            dk[cc] = s_kb[w][a * D + cc] - s_kb[w][bb * D + cc];
            rdk = fma(rho[cc], dk[cc], rdk);
          }
          qq *= ri * ri;
          const double tq = tr - qq;                       // sum_i (|delta_i|^2 - r1_i^2)
          const double F2 = fma(f2, qq, fma(f1ri, tq, f1ri * rdk)), F1x2 = 2.0 * f1ri;
          double* rec = &s_rec[id >> 8];
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            rec[cc] = rho[cc];
            rec[QPW + cc] = fma(F2, rho[cc], fma(F1x2, w1[cc], f0 * dk[cc]));    // D_v[kbar] + the second-order source, this radius
          }
          rec[QF0] = f0; rec[QF1] = f1ri; rec[QGQ] = gq;
          dsum = fma(2.0, fma(f1, rr, D * f0), dsum);      // this radius' share of div v
          qsum += fma(Bc, qq, gq * tq);                    // ... and of the Laplacian source
        }
      }
      // the one-body radius of the lane's particle: mu(|z_a|) z_a with W = S_aa; both lanes of the particle hold the same numbers, the
      // lane of component 0 counts the radius' share of div v and of the Laplacian source
      double ob_f0 = 0.0, ob_fc = 0.0, ob_gq = 0.0, ob_pw = 0.0;
      {
        double hd[NH];
        if constexpr (TAB) {
          if (ob_ok) ff_table_eval<NH>(ob_T, ob_dr, hd);
          else {
#pragma unroll
            for (int m = 0; m < NH; m++) hd[m] = 0.0;
          }
        } else {
          ff_heads<NH, true>(s_w[1], s_e2, Hm, ob_r, hd);
        }
        if (ob_act) {
          const double f0 = hd[0], f1 = hd[1], f2 = hd[2], f3 = hd[3];
          const double Ac = fma(f2, ob_r, (1.0 + D) * f1), Bc = fma(f3, ob_r, (2.0 + D) * f2);
          const double f1ri = f1 * ob_ri, gq = Ac * ob_ri;
          const double* Sg = s_A[w];
          double Ws[D][D];
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
#pragma unroll
            for (int c2i = cc; c2i < D; c2i++) {
              const double ww = Sg[(ra * D + cc) * AS + ra * D + c2i];
              Ws[cc][c2i] = ww; Ws[c2i][cc] = ww;
            }
          }
          double w1[D], dk[D], qq = 0.0, tr = 0.0, rdk = 0.0;
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            double t = 0.0;
#pragma unroll
            for (int c2i = 0; c2i < D; c2i++) {
              const double ww = Ws[cc][c2i];
              t = fma(ww, ob_rho[c2i], t);
              if (c2i == cc) tr += ww;
            }
            w1[cc] = t;
            qq = fma(ob_rho[cc], t, qq);
            dk[cc] = s_kb[w][ra * D + cc];
            rdk = fma(ob_rho[cc], dk[cc], rdk);
          }
          qq *= ob_ri * ob_ri;
          const double tq = tr - qq;
          const double F2 = fma(f2, qq, fma(f1ri, tq, f1ri * rdk)), F1x2 = 2.0 * f1ri;
          double pw = fma(F2, ob_rho[0], fma(F1x2, w1[0], f0 * dk[0]));
#pragma unroll
          for (int cc = 1; cc < D; cc++) pw = (rc == cc) ? fma(F2, ob_rho[cc], fma(F1x2, w1[cc], f0 * dk[cc])) : pw;
          ob_f0 = f0; ob_fc = f1ri; ob_gq = gq; ob_pw = pw;
          if (rc == 0) {
            dsum += fma(f1, ob_r, D * f0);
            qsum += fma(Bc, qq, gq * tq);
          }
        }
      }
      FF_WG1_SYNC();      // records complete; S has been read: A takes its place
      nev++;
      FF_STAMP(2);
      FF_SCHED_FENCE();
      // ------------------------------------------------------------------ row lanes: own-row sums and row p of A
      double vi = 0.0, wk = 0.0, gdi = 0.0;
      {
        double Ad[D];
        double* arow = owner ? &s_A[w][p * AS] : s_dump;
        double dsel[D];      // row rc of the identity
#pragma unroll
        for (int cc = 0; cc < D; cc++) dsel[cc] = rc == cc ? 1.0 : 0.0;
        {   // the one-body radius, from the lane's registers
          double ru = ob_rho[0];
#pragma unroll
          for (int cc = 1; cc < D; cc++) ru = (rc == cc) ? ob_rho[cc] : ru;
          const double fc = ob_fc * ru;
          vi = ob_f0 * ru;
          wk = ob_pw;
          gdi = ob_gq * ru;
#pragma unroll
          for (int cc = 0; cc < D; cc++) Ad[cc] = fma(fc, ob_rho[cc], ob_f0 * dsel[cc]);
        }
#pragma unroll
        for (int j = 0; j < N - 1; j++) {
          int pk = prec[j];
          FF_OPAQUE(pk);
          const double* rec = &s_rec[(pk >> 1) & 0xfff];
          const int smask = (int)((unsigned)pk << 31);                                        // the sign of this partner's odd terms, as a sign bit
          // (rho is read once, both components, and the lane's own one selected: rec[rc] beside rec[0 .. D) was a seventh LDS read per partner)
          double rh[D];
#pragma unroll
          for (int cc = 0; cc < D; cc++) rh[cc] = rec[cc];
          double ru = rh[0];
#pragma unroll
          for (int cc = 1; cc < D; cc++) ru = (rc == cc) ? rh[cc] : ru;
          const double f0 = rec[QF0], fc = rec[QF1] * ru;
          const double rcv = __hiloint2double(__double2hiint(ru) ^ smask, __double2loint(ru));
          const double pw = rec[QPW + rc];
          vi = fma(f0, rcv, vi);
          wk += __hiloint2double(__double2hiint(pw) ^ smask, __double2loint(pw));
          gdi = fma(rec[QGQ], rcv, gdi);
          double* ablk = (double*)((char*)arow + (pk >> 13));       // block k(j) of the lane's row
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            const double Bcc = fma(fc, rh[cc], f0 * dsel[cc]);               // B = eta I + (eta'/r) rho rho^T, row rc
            Ad[cc] += Bcc;
            ablk[cc] = -Bcc;
          }
        }
#pragma unroll
        for (int cc = 0; cc < D; cc++) arow[ra * D + cc] = Ad[cc];
        (owner ? &s_gd[w][p] : &s_dump[MP])[0] = gdi;
      }
      FF_WG1_SYNC();
      FF_STAMP(3);
      FF_SCHED_FENCE();
      // ------------------------------------------------------------------ J' = A J on the matrix cores, (grad Delta)' = -J^T g
      {
        double Jin[NB];
        if (use_y) {
#pragma unroll
#ifdef FF_REFORM_JIN
          for (int e = 0; e < NB; e++) Jin[e] = form(1 + e);
#else
          for (int e = 0; e < NB; e++) Jin[e] = Jk[e];
#endif
        } else {
#pragma unroll
          for (int e = 0; e < NB; e++) Jin[e] = form_noy(1 + e);
        }
        const double* Aw = s_A[w];
#pragma unroll
        for (int I = 0; I < MB; I++) {
          double Aop[MB];
#pragma unroll
          for (int K = 0; K < MB; K++) Aop[K] = Aw[(4 * K + r) * AS + 4 * I + c];
#pragma unroll
          for (int Kc = 0; Kc < MB; Kc++) {
            double acc = 0.0;
#pragma unroll
            for (int K = 0; K < MB; K++) acc = ff_mfma4(Aop[K], Jin[K * MB + Kc], acc);
            out[1 + I * MB + Kc] = acc;
          }
        }
        // Component 4K + c of g^T J is a sum over the rows 4I + r -- itself a product on the matrix cores: with A_blk[i][k] = g[4I + k] for
        // every i (lane (r, c) supplies g[4I + r]: the value it reads from LDS anyway) and B = block (I, K) of J, the sum over I of the
        // products is (g^T J)[4K + j] on every lane (i, j); the lane keeps the component of its own coordinate p = 4r + c, i.e. K = r.
        // MB^2 matrix instructions instead of MB^2 fp64 FMAs and a reduce-scatter through the LDS crossbar (round 6: three exchanges).
        // The two scalars -- Delta' = -div v and (lap Delta)' = -(sum_i D2div[u_i, u_i] + g . kbar) -- are summed over the quad by DPP
        // and over the walker's four quads by one product with ones: column 0 carries the first, the other columns the second; the
        // spare lanes p = 12, 13 (r = 3, c = 0, 1) integrate them.
        double gdr[MB];
#pragma unroll
        for (int I = 0; I < MB; I++) gdr[I] = s_gd[w][4 * I + r];
        double dd = 0.0;
#pragma unroll
        for (int K = 0; K < MB; K++) {
          double t = 0.0;
#pragma unroll
          for (int I = 0; I < MB; I++) t = ff_mfma4(gdr[I], Jin[I * MB + K], t);
          dd = (r == K) ? t : dd;
        }
        double res;
        {
          double ds4 = dsum, qs4 = fma(gdi, kin, qsum);
          ds4 += ff_swap1(ds4); ds4 += ff_swap2(ds4);
          qs4 += ff_swap1(qs4); qs4 += ff_swap2(qs4);
          res = quads_sum(c == 0 ? ds4 : qs4);
        }
        out[0] = owner ? vi : ((p == PDL || p == PLP) ? -res : 0.0);
        out[IK] = wk;                         // (row lanes beyond M read the zero record: all their sums vanish)
        out[IDD] = -dd;
      }
      FF_STAMP(4);
      FF_SCHED_FENCE();
      // ------------------------------------------------------------------ consume (Dormand-Prince bookkeeping)
#if defined(FF_STAMPS) && defined(FF_STAMPS_TRACE)
      const int s_prev = SG == FF_STAGE_DYN ? s : SG;
#endif
      if (at(-2)) {
        ctl_t C; C.get(s_ctl[w]);
        FF_WG1_SYNC();
        const double sens_w = C.sens_w, w0 = owner ? 1.0 : A.sum_w * sens_w;
        auto wgt = [&](int v) -> double { return v == 0 ? w0 : sens_w; };
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double yv = y.get(v);
          const double isc = wgt(v) * ff_rcp(fma(fabs(yv), rtol, atol));
          p0 = fma(yv * isc, yv * isc, p0);
          p1 = fma(c0[v] * isc, c0[v] * isc, p1);
        }
        const double d0 = sqrt(walker_sum(p0) * (1.0 / NT));
        C.d1v = sqrt(walker_sum(p1) * (1.0 / NT));
        C.h0v = C.S.h0(d0, C.d1v);
        s = -1;
        if (!ff_wave_or(&s_any, lane, (!C.S.done && !(C.hwarm > 0.0)) ? 1 : 0)) {   // every walker of the wave brings its own first step
          C.S.habs = fmin(C.hwarm, C.S.interval);
          C.S.plan();
          s = 1;
        }
        hs_keep = C.S.h;
        s_ctl[w].get(C);
        FF_WG1_SYNC();
      } else if (at(-1)) {
        ctl_t C; C.get(s_ctl[w]);
        FF_WG1_SYNC();
        const double sens_w = C.sens_w, w0 = owner ? 1.0 : A.sum_w * sens_w;
        auto wgt = [&](int v) -> double { return v == 0 ? w0 : sens_w; };
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double yv = y.get(v);
          const double t = (out[v] - c0[v]) * wgt(v) * ff_rcp(fma(fabs(yv), rtol, atol));
          p2 = fma(t, t, p2);
        }
        const double d2 = sqrt(walker_sum(p2) * (1.0 / NT)) / C.h0v;
        C.S.init_habs(C.h0v, C.d1v, d2);
        if (C.hwarm > 0.0) C.S.habs = fmin(C.hwarm, C.S.interval);
        C.S.plan();
        hs_keep = C.S.h;
        s_ctl[w].get(C);
        FF_WG1_SYNC();
        s = 1;
      } else if (at(0)) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        s = 1;
      } else if (at(1)) {
#pragma unroll
        for (int v = 0; v < NV; v++) c1[v] = out[v];
        s = 2;
      } else if (at(2)) {
#pragma unroll
        for (int v = 0; v < NV; v++) c2[v] = out[v];
        s = 3;
      } else if (at(3)) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double k0v = c0[v], k1v = c1[v], k2v = c2[v], k3v = out[v], yv = y.get(v);
          c0[v] = fma(hs, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, yv);
          c1[v] = fma(hs, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, yv);
          c2[v] = fma(hs, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, yv);
          c3.set(v, hs * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v));
          // (with the stages laid out one behind the other the compiler sinks these sums to where they are used -- c1 into stage 5, c2
          // into stage 6 -- and keeps k0 .. k3, 96 registers, alive through the right-hand sides in between: 112 B of scratch per
          // lane.  An empty asm pins each result here.)
          FF_OPAQUE(c0[v]); FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]);
        }
        s = 4;
      } else if (at(4)) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c1[v] = fma(hs * FF_A54, out[v], c1[v]);
          c2[v] = fma(hs * FF_B4, out[v], c2[v]);
          c0[v] = fma(hs * FF_E4, out[v], c3.get(v));      // c0 -- the input of this stage -- is free now: the error accumulator lives there
          FF_OPAQUE(c0[v]); FF_OPAQUE(c1[v]); FF_OPAQUE(c2[v]);
        }                                                   // through stages 5 and 6 (its LDS columns: one write at stage 3, one read here)
        s = 5;
      } else if (at(5)) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c2[v] = fma(hs * FF_B5, out[v], c2[v]);
          c0[v] = fma(hs * FF_E5, out[v], c0[v]);
          FF_OPAQUE(c0[v]); FF_OPAQUE(c2[v]);
        }
        s = 6;
      } else {
        const double sens_w = s_ctl[w].sens_w, w0 = owner ? 1.0 : A.sum_w * sens_w;
        auto wgt = [&](int v) -> double { return v == 0 ? w0 : sens_w; };
        double pe = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double e = fma(hs * FF_E6, out[v], c0[v]), yv = y.get(v);
          const double t = e * wgt(v) * ff_rcp(fma(fmax(fabs(yv), fabs(c2[v])), rtol, atol));   // c2 = the candidate y_new
          pe = fma(t, t, pe);
        }
        const double err = sqrt(walker_sum(pe) * (1.0 / NT));
        ctl_t C; C.get(s_ctl[w]);
        FF_WG1_SYNC();
        const bool was_active = !C.S.done;
        const bool acc = C.S.decide(err, A.max_steps);
        if (acc) C.hmax_acc = fmax(C.hmax_acc, fabs(hs));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y.set(v, c2[v]); c0[v] = out[v]; }
        }
        C.S.plan();
        hs_keep = C.S.h;
        s_ctl[w].get(C);
        FF_WG1_SYNC();
        const int any = ff_wave_or(&s_any, lane, C.S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        s = (any & 2) ? 0 : 1;
        if (!any) return true;
      }
#if defined(FF_STAMPS) && defined(FF_STAMPS_TRACE)      // consume ticks by stage (the price list of a per-walker stage index: DESIGN.md 3o)
      { unsigned long long t_ = __builtin_amdgcn_s_memtime(); stage_acc[s_prev + 2] += t_ - stamp_prev; stage_cnt[s_prev + 2]++; }
#endif
      FF_STAMP(6);
      FF_SCHED_FENCE();
      return false;
    };
    // the evaluations in front of a step, then its six stages
#pragma unroll 1
    for (;;) {
#pragma unroll 1
      while (s <= 0) evaluate(ff_stage_c<FF_STAGE_DYN>{});
      evaluate(ff_stage_c<1>{});
      evaluate(ff_stage_c<2>{});
      evaluate(ff_stage_c<3>{});
      evaluate(ff_stage_c<4>{});
      evaluate(ff_stage_c<5>{});
      if (evaluate(ff_stage_c<6>{})) break;
    }
    // ---------------------------------------------------------------------- fused finish (ff_eloc, nup = ndn; wave-uniform branch)
    // What ff_eloc_slater_fixed_kernel + ff_eloc_contract_kernel did from the workspace, on the walker's sixteen lanes while J is
    // still on chip (src/utils.py:56-63, src/VMC.py:48-55):
    //   grad_x logp = J^T g0 - grad Delta                       g0 = grad_z logp0(z(t0)),  H0 = its Hessian
    //   lap_x logp  = tr(H0 S) + g0 . kbar - lap Delta          S = J J^T: the product the right-hand side forms on the matrix cores
    // -- sum_i u_i^T H0 u_i over the columns u_i of J IS tr(H0 J J^T), so no lane ever needs a column of J.  One lane per particle
    // builds its row of the Slater tables (Hermite functions and both derivatives from one recurrence pass; the 3 x 3, 2 x 2 or 1 x 1
    // inverse by the adjugate) and contracts it with S; J^T g0 is the transposed-block quad reduction of (grad Delta)' = -J^T g; the
    // potential comes from the radius lanes on x.  (ff_slater_fixed, the routine of the stand-alone finish, costs 55 us per 131 072
    // determinants: coefficient-table loads by a per-lane index and 256 registers -- inlined here it took 0.5 ms per launch.)
    bool fin_done = false;
    if constexpr (N % 2 == 0 && D == 2) {
      if (A.fin.on & 1) {
        constexpr int NSF = N / 2;
        fin_done = true;
        int ln = lane;
        FF_OPAQUE(ln);
        const int r = ln >> 4, w = (ln >> 2) & 3, c = ln & 3, p = 4 * r + c, tl = 16 * c + 4 * w + r;
        const bool owner = p < M;
        int64_t b;
        bool valid;
        walker_of(w, b, valid);
        {   // S = J J^T of the final J -> s_A[w]
          double Jt[NB];
          const double idn = r == c ? 1.0 : 0.0;
#pragma unroll
          for (int e = 0; e < NB; e++) Jt[e] = ff_block_transpose(y.get(1 + e), idn, tl);
          double* Sw = s_A[w];
#pragma unroll
          for (int I = 0; I < MB; I++) {
#pragma unroll
            for (int Kc = I; Kc < MB; Kc++) {
              double acc = 0.0;
#pragma unroll
              for (int K = 0; K < MB; K++) acc = ff_mfma4(Jt[I * MB + K], Jt[Kc * MB + K], acc);
              Sw[(4 * I + r) * AS + 4 * Kc + c] = acc;
              if (Kc != I) Sw[(4 * Kc + c) * AS + 4 * I + r] = acc;
            }
          }
        }
        const double y0 = y.get(0);                      // z_p (owner lanes), Delta (p = PDL), lap Delta (p = PLP)
        if (owner) s_z[w][p] = y0;
        FF_WG1_SYNC();
        // Particle lanes: lane i16 = 4r + c < N of a walker owns particle i16 (spin i16 / NSF): its row of the Slater matrix, of the
        // gradient tables T_c[a][b] = sum_j d_c phi_j(r_a) Dinv[j][b] and its same-particle Hessian sums S3; the rows
        // meet in LDS (the record area is free here), every lane inverts its spin's matrix itself (adjugate, <= 3 x 3).
        const int i16 = 4 * r + c;
        const bool pl = i16 < N;
        const int sp = pl ? i16 / NSF : 0, al = pl ? i16 - sp * NSF : 0, off = sp * NSF;
        constexpr int FS = 6 * NSF * NSF;                  // per walker: D [2][NSF][NSF] | T [2][2][NSF][NSF]
        double* const fD = s_rec + w * FS + sp * NSF * NSF;
        double* const fT = s_rec + w * FS + 2 * NSF * NSF + sp * 2 * NSF * NSF;
        int deg[2 * NSF];
        {
          const int32_t* tab = sp ? A.fin.tab_dn : A.fin.tab_up;
          const int st = (A.fin.wstate && valid) ? A.fin.wstate[b] : 0;
#pragma unroll
          for (int j = 0; j < NSF; j++) ff_orb_decode(tab[st * NSF + j], deg[j], deg[NSF + j]);
        }
        int md = 1;
        {
          int mx = 0;
#pragma unroll
          for (int j = 0; j < 2 * NSF; j++) mx = deg[j] > mx ? deg[j] : mx;
          for (int dgr = 7; dgr >= 2; dgr--)
            if (__ballot(pl && mx >= dgr)) { md = dgr; break; }
        }
        const double zx = s_z[w][pl ? 2 * i16 : 0], zy = s_z[w][pl ? 2 * i16 + 1 : 1];
        double ph[NSF], pgx[NSF], pgy[NSF], pxx[NSF], pxy[NSF], pyy[NSF];
        {
          double hx[NSF], hx1[NSF], hx2[NSF], hy[NSF], hy1[NSF], hy2[NSF];
          ff_herm_rec_d2<NSF>(deg, zx, md, hx, hx1, hx2);
          ff_herm_rec_d2<NSF>(deg + NSF, zy, md, hy, hy1, hy2);
          const double gs = ff_gauss2d_fast(zx, zy);
#pragma unroll
          for (int j = 0; j < NSF; j++) {      // psi(x) = e^{-x^2/2} h(x): psi' = h' - x h, psi'' = h'' - 2 x h' + (x^2 - 1) h  (times the Gaussian)
            const double px1 = fma(-zx, hx[j], hx1[j]), py1 = fma(-zy, hy[j], hy1[j]);
            const double px2 = fma(fma(zx, zx, -1.0), hx[j], fma(-2.0 * zx, hx1[j], hx2[j]));
            const double py2 = fma(fma(zy, zy, -1.0), hy[j], fma(-2.0 * zy, hy1[j], hy2[j]));
            const double gx = gs * hx[j], gy = gs * hy[j];
            ph[j] = gx * hy[j];
            pgx[j] = gs * px1 * hy[j]; pgy[j] = gx * py1;
            pxx[j] = gs * px2 * hy[j]; pxy[j] = gs * px1 * py1; pyy[j] = gx * py2;
          }
        }
        if (pl) {
#pragma unroll
          for (int j = 0; j < NSF; j++) fD[al * NSF + j] = ph[j];
        }
        FF_WG1_SYNC();
        double lp0 = 0.0, trhs = 0.0;
        double Tx[NSF], Ty[NSF], S3[3] = {0.0, 0.0, 0.0};
        {
          double Dm[NSF][NSF], Di[NSF][NSF];
#pragma unroll
          for (int a2 = 0; a2 < NSF; a2++)
#pragma unroll
            for (int j = 0; j < NSF; j++) Dm[a2][j] = fD[a2 * NSF + j];
          const double det = ff_inv_small<NSF>(Dm, Di);
          lp0 = (pl && al == 0) ? 2.0 * ff_log(fabs(det)) : 0.0;
#pragma unroll
          for (int bb2 = 0; bb2 < NSF; bb2++) {
            double tx = 0.0, ty = 0.0;
#pragma unroll
            for (int j = 0; j < NSF; j++) { tx = fma(pgx[j], Di[j][bb2], tx); ty = fma(pgy[j], Di[j][bb2], ty); }
            Tx[bb2] = tx; Ty[bb2] = ty;
          }
#pragma unroll
          for (int j = 0; j < NSF; j++) {
            double dja = Di[j][0];        // Dinv[j][al]: static select (no runtime-indexed register array)
#pragma unroll
            for (int a2 = 1; a2 < NSF; a2++) dja = (al == a2) ? Di[j][a2] : dja;
            S3[0] = fma(pxx[j], dja, S3[0]); S3[1] = fma(pxy[j], dja, S3[1]); S3[2] = fma(pyy[j], dja, S3[2]);
          }
        }
        if (pl) {
          double txa = Tx[0], tya = Ty[0];
#pragma unroll
          for (int a2 = 1; a2 < NSF; a2++) { txa = (al == a2) ? Tx[a2] : txa; tya = (al == a2) ? Ty[a2] : tya; }
          s_kb[w][2 * i16] = 2.0 * txa;           // g0 = grad_z logp0: 2 T_c[a][a]
          s_kb[w][2 * i16 + 1] = 2.0 * tya;
#pragma unroll
          for (int bb2 = 0; bb2 < NSF; bb2++) { fT[al * NSF + bb2] = Tx[bb2]; fT[NSF * NSF + al * NSF + bb2] = Ty[bb2]; }
        }
        FF_WG1_SYNC();
        if (pl) {     // this particle's share of tr(H0 S): same-particle block with S3, cross blocks -T_ac (x) T_ca
          const double* Sm = s_A[w];
          const int ia = 2 * i16;
          double q = S3[0] * Sm[ia * AS + ia] + 2.0 * S3[1] * Sm[ia * AS + ia + 1] + S3[2] * Sm[(ia + 1) * AS + ia + 1];
#pragma unroll
          for (int cc = 0; cc < NSF; cc++) {
            const int ic = 2 * (off + cc);
            const double tcx = fT[cc * NSF + al], tcy = fT[NSF * NSF + cc * NSF + al];
            q -= Tx[cc] * (tcx * Sm[ia * AS + ic] + tcy * Sm[ia * AS + ic + 1]) + Ty[cc] * (tcx * Sm[(ia + 1) * AS + ic] + tcy * Sm[(ia + 1) * AS + ic + 1]);
          }
          trhs = 2.0 * q;
        }
        FF_WG1_SYNC();
        // grad_x logp: component p of J^T g0 (rows 4I + r on the lane, r across the walker's quads), minus grad Delta
        double dd = 0.0;
#pragma unroll
        for (int K = 0; K < MB; K++) {
          double t = 0.0;
#pragma unroll
          for (int I = 0; I < MB; I++) t = ff_mfma4(s_kb[w][4 * I + r], y.get(1 + I * MB + K), t);      // (as (grad Delta)' in the right-hand side)
          dd = (r == K) ? t : dd;
        }
        const double g0p = owner ? s_kb[w][p] : 0.0;
        const double gradp = owner ? dd - y.get(IDD) : 0.0;
        const double lapv = walker_sum(trhs + (owner ? g0p * y.get(IK) : 0.0) - (p == PLP ? y0 : 0.0));
        const double logpv = walker_sum(lp0 - (p == PDL ? y0 : 0.0));
        const double g2 = walker_sum(gradp * gradp);
        // V(x) = sum_{i<j} Z / r_ij + (1/2) sum_i r_i^2 on the radius lanes (src/potentials.py:13, 23-47)
        const double xp = ff_opt_load(A.y_in, valid && owner, b * M + p, A.y_in, 0.25 * (p + 1) + 0.125 * ((p * 7) % 5));
        FF_WG1_SYNC();
        if (owner) s_z[w][p] = xp;
        FF_WG1_SYNC();
        double vl = 0.0;
#pragma unroll
        for (int qk = 0; qk < NQ; qk++) {
          int id = rq_id[qk];
          FF_OPAQUE(id);
          const bool act = id >= 0;
          const int a = act ? (id & 15) : 0, bb = act ? ((id >> 4) & 15) : 1;
          double r2 = 0.0;
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            const double d = s_z[w][a * D + cc] - s_z[w][bb * D + cc];
            r2 = fma(d, d, r2);
          }
          double rr, ri;
          ff_sqrt_rcp(r2, rr, ri);
          vl += act ? A.fin.Z * ri : 0.0;
        }
        // (the trap term is taken by the coordinate lanes)
        if (A.fin.use_ho && owner) vl = fma(0.5 * xp, xp, vl);
        const double Vv = walker_sum(vl);
        if (valid) {
          ctl_t C; C.get(s_ctl[w]);
          const ff_stepper& S = C.S;
          const bool failed = S.fail != 0;
          const double bad = failed ? __builtin_nan("") : 0.0;
          if (owner) {
            A.y_out[b * M + p] = y0 + bad;
            if (A.fin.grad) A.fin.grad[b * M + p] = gradp + bad;
            if (A.fin.glogp0) A.fin.glogp0[b * M + p] = g0p + bad;
          }
          if (p == PDL) {
            A.dl_out[b] = y0 + bad;
            if (A.fin.logp) A.fin.logp[b] = logpv + bad;
            if (A.fin.lap) A.fin.lap[b] = lapv + bad;
            if (A.fin.V) A.fin.V[b] = Vv;
            if (A.fin.eloc) A.fin.eloc[b] = -0.25 * lapv - 0.125 * g2 + Vv + bad;
          }
          if (r == 0 && c == 0) {
            if (A.h_out) A.h_out[b] = C.hmax_acc > 0.0 ? C.hmax_acc : C.hwarm;
            if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
            if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
          }
        }
      }
    }
    // ---------------------------------------------------------------------- results
    if (!fin_done) {
     int ln = lane;
     FF_OPAQUE(ln);
     const int r = ln >> 4, w = (ln >> 2) & 3, c = ln & 3, p = 4 * r + c;
     const bool owner = p < M;
     int64_t b;
     bool valid;
     walker_of(w, b, valid);
     if (valid) {
      ctl_t C; C.get(s_ctl[w]);
      const ff_stepper& S = C.S;
      const double hmax_acc = C.hmax_acc, hwarm = C.hwarm;
      const bool failed = S.fail != 0;
      const double bad = failed ? __builtin_nan("") : 0.0;   // failed integration -> NaN results (see ff_ode_fwd_kernel)
      if (owner) {
        A.y_out[b * M + p] = y.get(0) + bad;
        A.kbar[b * M + p] = y.get(IK);
        A.dD[b * M + p] = y.get(IDD);
        if (p > 0) A.Lpart[b * M + p] = 0.0;
      }
      if (p == PDL) A.dl_out[b] = y.get(0) + bad;
      if (p == PLP) A.Lpart[b * M] = y.get(0);
#pragma unroll
      for (int I = 0; I < MB; I++)
#pragma unroll
        for (int K = 0; K < MB; K++) {
          if (4 * I + r < M && 4 * K + c < M) A.Jt[(b * M + 4 * K + c) * M + 4 * I + r] = y.get(1 + I * MB + K);   // Jt[b][i][k] = dz_k/dx_i
        }
      if (r == 0 && c == 0) {
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
      }
     }
    }
    FF_WG1_SYNC();
  }
#ifdef FF_STAMPS
  FF_STAMP(7);
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
#ifdef FF_STAMPS_TRACE
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) {
      atomicAdd((unsigned long long*)(A.stats + 65600) + q, stage_acc[q]);
      atomicAdd((unsigned long long*)(A.stats + 65600) + 9 + q, stage_cnt[q]);
    }
#endif
#endif
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  FF_WG1_SYNC();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}
