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
//   S  block (I, Kc)  = sum_K  mfma(Jt[I][K], Jt[Kc][K]),  Jt = the blocks transposed in place (lane (r,c) <-> (c,r): two
//                       ds_bpermute per double); only I <= Kc is computed, both halves are written to LDS.
// Everything else is per-radius or per-coordinate work:
//   R1  radius lanes (the wave's 4 R radii dealt over the lanes): heads from the radial table, then deposits -- the D x D
//       block B = eta I + (eta'/r) rho rho^T into A (off-diagonal particle blocks stored, diagonal ones accumulated with
//       LDS atomics), and each particle's velocity, A kbar and grad div sums (LDS atomics on a single wave: fixed order);
//   R2  radius lanes: W from S, the second-order sources, deposited the same way;
//   coordinate p = 4r + c (< M) is integrated by lane (r, c): z_p, kbar_p, its part of Delta and lap Delta; the component
//   4c + r of grad Delta sits on lane (r, c) (it comes out of the transposed blocks by a quad reduction).
#pragma once

template <int N, int D, bool TAB, int WPS = 1>
__global__ void __launch_bounds__(FF_WAVE, WPS)
ff_eloc_mfma_kernel(ff_fwd_args A) {
  constexpr int M = N * D, MB = (M + 3) / 4, MP = 4 * MB, G = 4;
  static_assert(MB >= 1 && MB <= 3, "at most 12 coordinates per walker");
  constexpr int P = N * (N - 1) / 2, R = P + N;
  constexpr int NH = 4, NB = MB * MB, NV = NB + 5;
  constexpr int IK = NB + 1, IDD = NB + 2, IDL = NB + 3, ILP = NB + 4;

  __shared__ ff_wtab s_w[TAB ? 1 : 2][TAB ? 1 : FF_HPAD];
  __shared__ double s_e2[TAB ? 1 : 64];
  constexpr int AS = MP + 2;     // row stride of the per-walker matrices (even: 16-byte aligned rows)
  __shared__ __attribute__((aligned(16))) double s_A[G][MP * AS], s_S[G][MP * AS];
  // per coordinate: z, kbar (published), v, A kbar + second-order source, grad div (accumulated); per walker: div v, lap source
  __shared__ double s_z[G][MP], s_kb[G][MP], s_v[G][MP], s_ws[G][MP], s_gd[G][MP], s_ds[G], s_ls[G];
  // records R1 -> R2: rho (D), eta'/r, c phi'/r, 1/r^2, eta'', c phi''
  constexpr int RW = (D + 5 + 1) & ~1;
  __shared__ __attribute__((aligned(16))) double s_rec[G * R * RW];
  __shared__ double s_err[FF_WAVE];
  __shared__ double s_cv[NV][FF_WAVE];   // error accumulator of the Dormand-Prince step, lane-private columns
  __shared__ int s_pa[R], s_pb[R], s_any;
  __shared__ int s_st[4];
  __shared__ long long s_next;

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
  if (lane == 0) {
    int q = 0;
    for (int a = 0; a < N; a++)
      for (int b = a + 1; b < N; b++) { s_pa[q] = a; s_pb[q] = b; q++; }
    for (int a = 0; a < N; a++) { s_pa[P + a] = a; s_pb[P + a] = -1; }
  }
  for (int e = lane; e < G * MP * AS; e += FF_WAVE) { (&s_A[0][0])[e] = 0.0; (&s_S[0][0])[e] = 0.0; }   // padding stays zero
  for (int e = lane; e < G * MP; e += FF_WAVE) { (&s_z[0][0])[e] = 0.0; (&s_kb[0][0])[e] = 0.0; (&s_gd[0][0])[e] = 0.0; }
  __syncthreads();
  const int He = A.net.He, Hm = A.net.Hm;
  const bool has_mu = Hm > 0;
  const int nrad = has_mu ? R : P;
  const double tab_inv_h = TAB ? rtab[0] : 0.0, tab_h = TAB ? rtab[1] : 0.0;
  const double rtol = A.rtol, atol = A.atol;
  constexpr double NT = (double)M * M + 4.0 * M + 2.0;   // z, J, kbar, grad Delta, Delta, lap Delta
  const int64_t ngroups = (A.B + G - 1) / G;
  // radii this lane evaluates (slot qk: radius lane + 64 qk of the wave's G*nrad): walker | a << 4 | b (15: none) << 8 | index << 12
  constexpr int NQ = (G * R + FF_WAVE - 1) / FF_WAVE;
  int rq_id[NQ];
#pragma unroll
  for (int qk = 0; qk < NQ; qk++) {
    const int q = lane + qk * FF_WAVE;
    const bool act = q < G * nrad;
    const int qg = act ? q / nrad : 0, pr = act ? q - qg * nrad : 0;
    rq_id[qk] = act ? (qg | (s_pa[pr] << 4) | ((s_pb[pr] < 0 ? 15 : s_pb[pr]) << 8) | (pr << 12)) : -1;
  }

  // sum over the 16 lanes of a walker (identical on all of them): quad by DPP, the four quads through LDS
  auto walker_sum = [&](double part) -> double {
    part += ff_swap1(part);
    part += ff_swap2(part);
    s_err[lane] = part;
    __syncthreads();
    const double t = (s_err[4 * w] + s_err[16 + 4 * w]) + (s_err[32 + 4 * w] + s_err[48 + 4 * w]);
    __syncthreads();
    return t;
  };

#ifdef FF_STAMPS
  unsigned long long stamp_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int64_t grp = blockIdx.x;; grp += gridDim.x) {
    if (A.queue) {   // persistent grid: next group from the launch's work counter (heavy walkers sit at the front)
      __syncthreads();
      if (lane == 0) s_next = (long long)atomicAdd(A.queue + (TAB ? 0 : 1), 1ULL);
      __syncthreads();
      grp = s_next;
    }
    if (grp >= ngroups) break;
    const int64_t bq = grp * G + w;
    const bool valid = bq < A.B;
    const int64_t b = ff_opt_load(A.order, valid, bq, A.y_in, (int32_t)bq);
    // Dormand-Prince storage as in ff_ode_fwd_kernel: y, c0..c2 (k0..k2, then the inputs of stages 4, 5 and y_new), c3 (error)
    double y[NV], c0[NV], c1[NV], c2[NV];
    ff_lane_vec<NV, true> c3(&s_cv[0][0], lane);
#pragma unroll
    for (int v = 0; v < NV; v++) { y[v] = 0.0; c0[v] = 0.0; c1[v] = 0.0; c2[v] = 0.0; c3[v] = 0.0; }
    {
      const double y0 = ff_opt_load(A.y_in, valid && owner, b * M + p, A.y_in, 0.25 * (p + 1) + 0.125 * ((p * 7) % 5));   // idle walkers: finite, distinct
      y[0] = owner ? y0 : y[0];
    }
#pragma unroll
    for (int I = 0; I < MB; I++) y[1 + I * MB + I] = (r == c && 4 * I + r < M) ? 1.0 : 0.0;   // J = identity
    ff_stepper S;
    S.begin(A.ta, A.tb, valid);
    // walkers of a low cost class: looser tolerance for the sensitivity components, larger first step (ff_ode.walker_class)
    const bool loose = ff_opt_load(A.wclass, valid, b, A.y_in, (int32_t)0x7fffffff) <= A.sens_class;
    const double hwarm = ff_opt_load(A.h_init, valid, A.h_scale < 0.0 ? 0 : b, A.y_in, 0.0) * (loose ? A.h_scale_loose : fabs(A.h_scale));
    const bool warm = hwarm > 0.0;
    // tolerance of the sensitivity components relative to the coordinates' (ff_ode.sens_tol): weight in the error norm
    const double sens_w = loose ? A.sens_w : 1.0;
    double hmax_acc = 0.0;
    int s = -2, nev = 0;
    double h0v = 0.0, d1v = 0.0;

#pragma unroll 1
    for (;;) {
      // ------------------------------------------------------------------ stage input (one expression for all stages)
      const double hs = S.h;
      double gy = 1.0, g0 = 0.0, g1 = 0.0, g2 = 0.0;
      switch (s) {
        case -1: g0 = h0v * S.dir; break;
        case 1: g0 = hs * FF_A10; break;
        case 2: g0 = hs * FF_A20; g1 = hs * FF_A21; break;
        case 3: g0 = hs * FF_A30; g1 = hs * FF_A31; g2 = hs * FF_A32; break;
        case 4: gy = 0.0; g0 = 1.0; break;
        case 5: gy = 0.0; g1 = 1.0; break;
        case 6: gy = 0.0; g2 = 1.0; break;
        default: break;   // -2, 0: the state itself
      }
      auto form = [&](int v) -> double { return fma(g2, c2[v], fma(g1, c1[v], fma(g0, c0[v], gy * y[v]))); };
      double out[NV];
      const double zin = form(0), kin = form(IK);
      FF_STAMP(0);
      // ------------------------------------------------------------------ publish z, kbar; clear the accumulators
      __syncthreads();
      if (owner) {
        s_z[w][p] = zin; s_kb[w][p] = kin;
        s_v[w][p] = 0.0; s_ws[w][p] = 0.0; s_gd[w][p] = 0.0;
      }
      if (r == 0 && c == 0) { s_ds[w] = 0.0; s_ls[w] = 0.0; }
      // the particle-diagonal D x D blocks of A are accumulated: clear this lane's elements of them
#pragma unroll
      for (int K = 0; K < MB; K++)
#pragma unroll
        for (int I = 0; I < MB; I++) {
          if (4 % D == 0 && K != I) continue;     // D = 2: a particle's rows never straddle two 4-blocks
          const int p1 = 4 * K + r, p2 = 4 * I + c;
          if (p1 / D == p2 / D && p1 < M && p2 < M) s_A[w][p1 * AS + p2] = 0.0;
        }
      __syncthreads();
      FF_STAMP(1);
      // ------------------------------------------------------------------ R1: radius lanes
      {
        double rq_rho[NQ][D], rq_dk[NQ][D], rq_r[NQ], rq_ri[NQ], rq_T[NQ][TAB ? NH + 5 : 1], rq_dr[NQ];
        bool rq_ok[NQ];
#pragma unroll
        for (int qk = 0; qk < NQ; qk++) {
          int id = rq_id[qk];
          FF_OPAQUE(id);
          const bool act = id >= 0;
          const int qg = act ? (id & 15) : 0, a = act ? ((id >> 4) & 15) : 0, bb0 = act ? ((id >> 8) & 15) : 15;
          const bool pair = bb0 != 15;
          const int bb = pair ? bb0 : a;
          double r2 = 0.0;
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            rq_rho[qk][cc] = s_z[qg][a * D + cc] - (pair ? s_z[qg][bb * D + cc] : 0.0);
            rq_dk[qk][cc] = s_kb[qg][a * D + cc] - (pair ? s_kb[qg][bb * D + cc] : 0.0);
            r2 = fma(rq_rho[qk][cc], rq_rho[qk][cc], r2);
          }
          ff_sqrt_rcp(r2, rq_r[qk], rq_ri[qk]);
          rq_dr[qk] = 0.0;
          rq_ok[qk] = true;
          if constexpr (TAB) {
            rq_ok[qk] = ff_table_fetch<NH>(rtab, tab_inv_h, tab_h, pair ? 0 : 1, rq_r[qk], rq_T[qk], rq_dr[qk]);
            if (act && !rq_ok[qk]) off_table = true;
          }
        }
#pragma unroll
        for (int qk = 0; qk < NQ; qk++) {
          int id = rq_id[qk];
          FF_OPAQUE(id);
          if (id >= 0) {
            const int qg = id & 15, a = (id >> 4) & 15, bb0 = (id >> 8) & 15, pr = id >> 12;
            const bool pair = bb0 != 15;
            const int bb = pair ? bb0 : a;
            const double* rho = rq_rho[qk];
            const double* dk = rq_dk[qk];
            const double rr = rq_r[qk], ri = rq_ri[qk];
            double hd[NH];
            if constexpr (TAB) {
              if (rq_ok[qk]) ff_table_eval<NH>(rq_T[qk], rq_dr[qk], hd);
              else {
#pragma unroll
                for (int m = 0; m < NH; m++) hd[m] = 0.0;
              }
            } else {
              ff_heads<NH, true>(s_w[pair ? 0 : 1], s_e2, pair ? He : Hm, rr, hd);
            }
            const double cf = pair ? 2.0 : 1.0;
            const double f0 = hd[0], f1 = hd[1], f2 = hd[2], f3 = hd[3];
            const double Ac = cf * fma(f2, rr, (1.0 + D) * f1), Bc = cf * fma(f3, rr, (2.0 + D) * f2);
            double rdk = 0.0;
#pragma unroll
            for (int cc = 0; cc < D; cc++) rdk = fma(rho[cc], dk[cc], rdk);
            const double f1ri = f1 * ri, F1k = f1ri * rdk, gq = Ac * ri;
            double* rec = &s_rec[(qg * R + pr) * RW];
#pragma unroll
            for (int cc = 0; cc < D; cc++) rec[cc] = rho[cc];
            rec[D] = f1ri; rec[D + 1] = gq; rec[D + 2] = ri * ri; rec[D + 3] = f2; rec[D + 4] = Bc;
            atomicAdd(&s_ds[qg], cf * fma(f1, rr, D * f0));       // this radius' share of div v
            double* Am = s_A[qg];
#pragma unroll
            for (int cc = 0; cc < D; cc++) {
              const double pv = f0 * rho[cc], pw = fma(F1k, rho[cc], f0 * dk[cc]), pg = gq * rho[cc];
              atomicAdd(&s_v[qg][a * D + cc], pv); atomicAdd(&s_ws[qg][a * D + cc], pw); atomicAdd(&s_gd[qg][a * D + cc], pg);
              if (pair) { atomicAdd(&s_v[qg][bb * D + cc], -pv); atomicAdd(&s_ws[qg][bb * D + cc], -pw); atomicAdd(&s_gd[qg][bb * D + cc], -pg); }
              const double fr = f1ri * rho[cc];
#pragma unroll
              for (int c2i = 0; c2i < D; c2i++) {
                const double Bcc = fma(fr, rho[c2i], c2i == cc ? f0 : 0.0);   // B = f0 I + (f1/r) rho rho^T
                atomicAdd(&Am[(a * D + cc) * AS + a * D + c2i], Bcc);         // dv_a/dz_a += B
                if (pair) {
                  atomicAdd(&Am[(bb * D + cc) * AS + bb * D + c2i], Bcc);     // dv_b/dz_b += B
                  Am[(a * D + cc) * AS + bb * D + c2i] = -Bcc;                // dv_a/dz_b = dv_b/dz_a = -B
                  Am[(bb * D + cc) * AS + a * D + c2i] = -Bcc;
                }
              }
            }
          }
        }
      }
      __syncthreads();
      nev++;
      FF_STAMP(2);
      // ------------------------------------------------------------------ the two products on the matrix cores
      {
        double Jin[NB], Jt[NB];
#pragma unroll
        for (int e = 0; e < NB; e++) { Jin[e] = form(1 + e); Jt[e] = ff_lane_read(Jin[e], tl); }
        double gdc[MB];   // grad div at the coordinates 4I + c (for the grad-Delta product with the transposed blocks)
#pragma unroll
        for (int I = 0; I < MB; I++) gdc[I] = s_gd[w][4 * I + c];
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
        double* Sw = s_S[w];
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
        // grad Delta' = -J^T g: component 4K + r (K = c) from the transposed blocks, summed over the quad
        double dd = 0.0;
#pragma unroll
        for (int K = 0; K < MB; K++) {
          double t = 0.0;
#pragma unroll
          for (int I = 0; I < MB; I++) t = fma(gdc[I], Jt[I * MB + K], t);
          t += ff_swap1(t);
          t += ff_swap2(t);
          dd = (c == K) ? t : dd;
        }
        out[IDD] = -dd;
      }
      __syncthreads();
      FF_STAMP(3);
      // ------------------------------------------------------------------ R2: radius lanes contract their terms with S
#pragma unroll
      for (int qk = 0; qk < NQ; qk++) {
        int id = rq_id[qk];
        FF_OPAQUE(id);
        if (id >= 0) {
          const int qg = id & 15, a = (id >> 4) & 15, bb0 = (id >> 8) & 15, pr = id >> 12;
          const bool pair = bb0 != 15;
          const int bb = pair ? bb0 : a;
          const double* rec = &s_rec[(qg * R + pr) * RW];
          double rho[D];
#pragma unroll
          for (int cc = 0; cc < D; cc++) rho[cc] = rec[cc];
          const double f1ri = rec[D], gq = rec[D + 1], ri2 = rec[D + 2], f2 = rec[D + 3], Bc = rec[D + 4];
          const double* Sg = s_S[qg];
          double W[D][D];
#pragma unroll
          for (int cc = 0; cc < D; cc++)
#pragma unroll
            for (int c2i = 0; c2i < D; c2i++) {
              double ww = Sg[(a * D + cc) * AS + a * D + c2i];
              if (pair) ww += Sg[(bb * D + cc) * AS + bb * D + c2i] - Sg[(a * D + cc) * AS + bb * D + c2i] - Sg[(a * D + c2i) * AS + bb * D + cc];
              W[cc][c2i] = ww;
            }
          double w1[D], qq = 0.0, tr = 0.0;
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            double t = 0.0;
#pragma unroll
            for (int c2i = 0; c2i < D; c2i++) t = fma(W[cc][c2i], rho[c2i], t);
            w1[cc] = t;
            qq = fma(rho[cc], t, qq);
            tr += W[cc][cc];
          }
          qq *= ri2;
          const double tq = tr - qq;                       // sum_i (|delta_i|^2 - r1_i^2)
          const double F2 = fma(f2, qq, f1ri * tq), F1x2 = 2.0 * f1ri;
#pragma unroll
          for (int cc = 0; cc < D; cc++) {
            const double quad = fma(F2, rho[cc], F1x2 * w1[cc]);
            atomicAdd(&s_ws[qg][a * D + cc], quad);
            if (pair) atomicAdd(&s_ws[qg][bb * D + cc], -quad);
          }
          atomicAdd(&s_ls[qg], fma(Bc, qq, gq * tq));
        }
      }
      __syncthreads();
      FF_STAMP(4);
      // ------------------------------------------------------------------ per-coordinate sums
      {
        const double gdp = owner ? s_gd[w][p] : 0.0;
        out[0] = owner ? s_v[w][p] : 0.0;
        out[IK] = owner ? s_ws[w][p] : 0.0;
        out[IDL] = (p == 0) ? -s_ds[w] : 0.0;
        out[ILP] = owner ? -(((p == 0) ? s_ls[w] : 0.0) + gdp * kin) : 0.0;
      }
      FF_STAMP(5);
      // ------------------------------------------------------------------ consume (Dormand-Prince bookkeeping)
      if (s == -2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double isc = (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p0 = fma(y[v] * isc, y[v] * isc, p0);
          p1 = fma(c0[v] * isc, c0[v] * isc, p1);
        }
        const double d0 = sqrt(walker_sum(p0) * (1.0 / NT));
        d1v = sqrt(walker_sum(p1) * (1.0 / NT));
        h0v = S.h0(d0, d1v);
        s = -1;
        if (!ff_wave_or(&s_any, lane, (!S.done && !warm) ? 1 : 0)) {   // every walker of the wave brings its own first step
          S.habs = fmin(hwarm, S.interval);
          S.plan();
          s = 1;
        }
      } else if (s == -1) {
        double p2 = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double t = (out[v] - c0[v]) * (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fabs(y[v]), rtol, atol));
          p2 = fma(t, t, p2);
        }
        const double d2 = sqrt(walker_sum(p2) * (1.0 / NT)) / h0v;
        S.init_habs(h0v, d1v, d2);
        if (warm) S.habs = fmin(hwarm, S.interval);
        S.plan();
        s = 1;
      } else if (s == 0) {
#pragma unroll
        for (int v = 0; v < NV; v++) c0[v] = out[v];
        s = 1;
      } else if (s == 1) {
#pragma unroll
        for (int v = 0; v < NV; v++) c1[v] = out[v];
        s = 2;
      } else if (s == 2) {
#pragma unroll
        for (int v = 0; v < NV; v++) c2[v] = out[v];
        s = 3;
      } else if (s == 3) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double k0v = c0[v], k1v = c1[v], k2v = c2[v], k3v = out[v], yv = y[v];
          c0[v] = fma(hs, FF_A40 * k0v + FF_A41 * k1v + FF_A42 * k2v + FF_A43 * k3v, yv);
          c1[v] = fma(hs, FF_A50 * k0v + FF_A51 * k1v + FF_A52 * k2v + FF_A53 * k3v, yv);
          c2[v] = fma(hs, FF_B0 * k0v + FF_B2 * k2v + FF_B3 * k3v, yv);
          c3[v] = hs * (FF_E0 * k0v + FF_E2 * k2v + FF_E3 * k3v);
        }
        s = 4;
      } else if (s == 4) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c1[v] = fma(hs * FF_A54, out[v], c1[v]);
          c2[v] = fma(hs * FF_B4, out[v], c2[v]);
          c3[v] = fma(hs * FF_E4, out[v], c3[v]);
        }
        s = 5;
      } else if (s == 5) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
          c2[v] = fma(hs * FF_B5, out[v], c2[v]);
          c3[v] = fma(hs * FF_E5, out[v], c3[v]);
        }
        s = 6;
      } else {
        double pe = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double e = fma(hs * FF_E6, out[v], c3[v]);
          const double t = e * (v >= 1 ? sens_w : 1.0) * ff_rcp(fma(fmax(fabs(y[v]), fabs(c2[v])), rtol, atol));   // c2 = the candidate y_new
          pe = fma(t, t, pe);
        }
        const double err = sqrt(walker_sum(pe) * (1.0 / NT));
        const bool was_active = !S.done;
        const bool acc = S.decide(err, A.max_steps);
        if (acc) hmax_acc = fmax(hmax_acc, fabs(hs));
        if (acc) {
#pragma unroll
          for (int v = 0; v < NV; v++) { y[v] = c2[v]; c0[v] = out[v]; }
        }
        S.plan();
        const int any = ff_wave_or(&s_any, lane, S.done ? 0 : ((was_active && !acc) ? 3 : 1));
        if (!any) break;
        s = (any & 2) ? 0 : 1;
      }
      FF_STAMP(6);
    }
    // ---------------------------------------------------------------------- results
    const double delta = walker_sum(y[IDL]);
    if (valid) {
      const bool failed = S.fail != 0;
      const double bad = failed ? __builtin_nan("") : 0.0;   // failed integration -> NaN results (see ff_ode_fwd_kernel)
      if (owner) {
        A.y_out[b * M + p] = y[0] + bad;
        A.kbar[b * M + p] = y[IK];
        A.Lpart[b * M + p] = y[ILP];
      }
      if (4 * c + r < M && c < MB) A.dD[b * M + 4 * c + r] = y[IDD];
#pragma unroll
      for (int I = 0; I < MB; I++)
#pragma unroll
        for (int K = 0; K < MB; K++) {
          if (4 * I + r < M && 4 * K + c < M) A.Jt[(b * M + 4 * K + c) * M + 4 * I + r] = y[1 + I * MB + K];   // Jt[b][i][k] = dz_k/dx_i
        }
      if (r == 0 && c == 0) {
        A.dl_out[b] = delta + bad;
        if (A.h_out) A.h_out[b] = hmax_acc > 0.0 ? hmax_acc : hwarm;
        if (A.wcost) A.wcost[b] = S.nacc + S.nrej;
        if (A.stats) { atomicAdd(&s_st[0], nev); atomicMax(&s_st[1], S.nacc); atomicAdd(&s_st[2], S.nrej); if (failed) atomicMax(&s_st[3], 1); }
      }
    }
    __syncthreads();
  }
#ifdef FF_STAMPS
  FF_STAMP(7);
  if (A.stats && lane == 0)
    for (int q = 0; q < 9; q++) atomicAdd((unsigned long long*)(A.stats + 8) + q, stamp_acc[q]);
#endif
  if constexpr (TAB) { if (off_table) *A.evt = A.evt_id; }
  __syncthreads();
  if (A.stats && lane == 0 && (s_st[0] || s_st[3])) {
    atomicAdd(&A.stats[0], s_st[0]);
    atomicMax(&A.stats[1], s_st[1]);
    atomicAdd(&A.stats[2], s_st[2]);
    if (s_st[3]) atomicMax(&A.stats[3], 1);
  }
}
