// k_swl.hip -- K5 for byte-sized jobs (KSW_XBYTE: mate rescue of reads below 250 bases, lib/aln/mem_alnreg.c:433): ksw_u8
// (lib/aln/ksw.c:111-230) run AS the striped kernel it is.  The reference keeps 16 unsigned bytes in an SSE register and walks the
// query in slen = ceil(qlen / 16) stripes; here the 16 byte lanes of that register are 16 lanes of the wavefront (a DPP row), so a
// wavefront runs FOUR jobs side by side, a lane holds its byte of every stripe (H, E, Hmax: slen registers each), and a step of the
// reference's inner loop -- a dozen SSE instructions on 16 cells -- is a dozen vector instructions on 64 cells:
//   _mm_slli_si128(x, 1)      one DPP row shift (lane k takes lane k-1 of its row, lane 0 takes 0)
//   adds / subs_epu8          32-bit arithmetic clamped to [0, 255] (v_med3)
//   __max_16                  a rotate-and-max over the row
//   the lazy-F loop           the prefix maximum it converges to: four DPP steps across the row (swl_row)
// The wave-per-job kernel (k_sw.hip) spends ~300 vector instructions on a target row of one job (two prefix scans per 64 columns
// reproduce what the striped loop converges to); this one spends ~250 on a row of four.  On the hg38-like genome mate rescue is 2.3 M
// jobs per chunk -- reads inside repeat families, each tried against up to 50 windows that are diverged copies of it.
// Everything that decides the result is the reference's own sequence of operations: the saturation at 255 and the early exits
// (ksw.c:205,209), which H the next row's E opens from (before lazy-F: ksw.c:160-169), b[] (ksw.c:191-200), Hmax and the smallest query end (ksw.c:212-216), the second pass of ksw_align2 over the
// reversed prefixes with KSW_XSTOP (ksw.c:356-364).
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

#define SWL_ROW_ROR(n) (0x120 + (n))
// lane k of a 16-lane row takes lane k-1's value, lane 0 takes 0
__device__ __forceinline__ int swl_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false); }
__device__ __forceinline__ int swl_row_max(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(1), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(2), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(4), 0xf, 0xf, false); v = v > t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(8), 0xf, 0xf, false); v = v > t ? v : t;
	return v;
}
__device__ __forceinline__ int swl_row_min(int v)
{
	int t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(1), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(2), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(4), 0xf, 0xf, false); v = v < t ? v : t;
	t = __builtin_amdgcn_update_dpp(v, v, SWL_ROW_ROR(8), 0xf, 0xf, false); v = v < t ? v : t;
	return v;
}
__device__ __forceinline__ unsigned int swl_row_or(unsigned int v)
{
	v |= (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, SWL_ROW_ROR(1), 0xf, 0xf, false);
	v |= (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, SWL_ROW_ROR(2), 0xf, 0xf, false);
	v |= (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, SWL_ROW_ROR(4), 0xf, 0xf, false);
	v |= (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, SWL_ROW_ROR(8), 0xf, 0xf, false);
	return v;
}
__device__ __forceinline__ int swl_clamp(int v, int hi) { return v < 0 ? 0 : v > hi ? hi : v; }   // one v_med3_i32

// One target row of the four jobs of a wave: the striped main loop (ksw.c:147-171), then what the lazy-F loop (ksw.c:173-186) converges
// to.  The main loop carries F from stripe to stripe inside a byte lane, i.e. along the lane's own run of consecutive query columns
// [k * slen, (k + 1) * slen); lazy F is there to carry it across byte lanes: round after round it shifts F one byte up and sweeps the
// stripes until no byte of F exceeds its H - (o_ins + e_ins).  Its exit test is exact (a byte whose F does not beat H - oe has
// nothing to add that the main loop has not propagated already), so the H it leaves is max(H, F) with F the max-plus prefix scan of the
// whole row -- which costs the same whatever the data: each lane's best H - oe + column * e_ins comes out of the main loop, an exclusive
// maximum over the lower lanes of the row is four DPP steps, and one more sweep of the stripes applies it.  (The loop itself, row by
// row of the wave, took up to 16 rounds x slen steps on the diverged repeat copies mate rescue meets on an hg38-like genome.)
// Returns the lane's maximum of the main loop's H (what __max_16 reduces: F from another lane cannot exceed it, ksw.c:173).
// UNI: slen is a scalar (the same in every row of the wave).
template <int SL, bool UNI>
__device__ __forceinline__ int swl_row(int (&H)[SL], int (&E)[SL], const uint32_t (&prof)[SL], int tsh, int cap, int slen,
                                       int oe_del, int e_del, int oe_ins, int e_ins, int col0_e)
{
	int f = 0, mxv = 0, hlast = 0, gl = NEG_BIG, ce = col0_e;   // ce: (this lane's column) * e_ins
#pragma unroll
	for (int j = 0; j < SL; ++j) hlast = j == slen - 1 ? H[j] : hlast;
	int h = swl_shl1(hlast);
#pragma unroll
	for (int j = 0; j < SL; ++j) {
		if (j < slen) {
			const int s = (int)(int8_t)(prof[j] >> tsh);
			h = swl_clamp(h + s, cap);
			int e = E[j];
			h = h > e ? h : e;
			h = h > f ? h : f;
			mxv = mxv > h ? mxv : h;
			const int hold = H[j];
			H[j] = h;
			e -= e_del;
			int t = h - oe_del;
			e = e > t ? e : t;
			E[j] = e > 0 ? e : 0;
			f -= e_ins;
			t = h - oe_ins;
			f = f > t ? f : t;
			f = f > 0 ? f : 0;
			t += ce; gl = gl > t ? gl : t;
			ce += e_ins;
			h = hold;
		}
	}
	// F from the lower lanes of the row: exclusive prefix maximum of gl, then F(column c) = that - (c - 1) * e_ins
	int t;
	t = __builtin_amdgcn_update_dpp(NEG_BIG, gl, DPP_ROW_SHR(1), 0xf, 0xf, false); gl = gl > t ? gl : t;
	t = __builtin_amdgcn_update_dpp(NEG_BIG, gl, DPP_ROW_SHR(2), 0xf, 0xf, false); gl = gl > t ? gl : t;
	t = __builtin_amdgcn_update_dpp(NEG_BIG, gl, DPP_ROW_SHR(4), 0xf, 0xf, false); gl = gl > t ? gl : t;
	t = __builtin_amdgcn_update_dpp(NEG_BIG, gl, DPP_ROW_SHR(8), 0xf, 0xf, false); gl = gl > t ? gl : t;
	int v = __builtin_amdgcn_update_dpp(NEG_BIG, gl, DPP_ROW_SHR(1), 0xf, 0xf, false) - (col0_e - e_ins);
	if (__ballot(v > 0) != 0) {
#pragma unroll
		for (int j = 0; j < SL; ++j) {
			if (j < slen) { H[j] = H[j] > v ? H[j] : v; v -= e_ins; }
		}
	}
	return mxv;
}

// SL: stripes held in registers (queries of up to 16 * SL columns)
template <int SL>
__global__ void __launch_bounds__(256)
k_swl(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order, long long n,
      bsx_sw_res_t *res, unsigned long long *bscratch, int bcap)
{
	const int lane = wave_lane(), g = lane >> 4, k = lane & 15;
	const long long wave_id = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * (blockDim.x >> 6);
	unsigned long long *b = bscratch + ((size_t)wave_id * 4 + g) * (size_t)bcap;
	const int oe_del = sc.o_del + sc.e_del, oe_ins = sc.o_ins + sc.e_ins, e_del = sc.e_del, e_ins = sc.e_ins;
	for (long long q4 = wave_id; q4 * 4 < n; q4 += n_waves) {
		const long long jj = q4 * 4 + g;
		const bool valid = jj < n;
		const int job = valid ? order[jj] : 0;
		bsx_sw_job_t J = jobs[valid ? job : order[0]];
		const int8_t *mat = J.use_ct ? sc.ctmat : sc.gamat;
		int shift = 127, mx = 0;
		for (int a = 0; a < 25; ++a) { const int m = mat[a]; shift = m < shift ? m : shift; mx = m > mx ? m : mx; }
		shift = (int)(uint8_t)(256 - (int)(uint8_t)shift);   // ksw.c:84-88
		const int cap = 255 - shift;   // adds_epu8(h, S + shift) then subs_epu8(.., shift): h + s clamped to [0, 255 - shift]
		bsx_sw_res_t o;
		o.score = 0; o.te = -1; o.qe = -1; o.score2 = -1; o.te2 = -1; o.tb = -1; o.qb = -1;
		int r_score = 0, r_te = -1, r_qe = -1;
		bool run = valid;
		for (int pass = 0; pass < 2; ++pass) {
			// pass 0: the job as given.  pass 1 (ksw_align2's second call): the query prefix [0, qe] and the target prefix [0, te] reversed,
			// the rest of the target as it was, KSW_XSTOP at the first pass's score
			const int qlen = pass ? r_qe + 1 : J.qlen, tlen = J.tlen, te_rev = pass ? r_te : -1;
			const int xtra = pass ? (BSX_KSW_XSTOP | r_score) : J.xtra;
			const int minsc = (xtra & BSX_KSW_XSUBO) ? (xtra & 0xffff) : 0x10000;
			const int endsc = (xtra & BSX_KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
			const int slen = (qlen + 15) >> 4;
			const int slen_s = __builtin_amdgcn_readfirstlane(wave_max_i32(run ? slen : 0));
			const bool uni = __ballot(run && slen != slen_s) == 0;
			int H[SL], E[SL], Hm[SL];
			uint32_t prof[SL];   // this lane's column of stripe j against target bases 0..3: a signed byte each (a padding column scores 0)
#pragma unroll
			for (int j = 0; j < SL; ++j) {
				H[j] = E[j] = Hm[j] = 0;
				const int c = j + k * slen;   // the striped layout: byte k of stripe j is query column j + k * slen (ksw.c:93-99)
				int q = 5;
				if (run && j < slen && c < qlen) {
					q = pass ? reads[(long long)J.qoff + (long long)(r_qe - c) * J.qdir] : reads[(long long)J.qoff + (long long)c * J.qdir];
					if (J.qcomp) q = q < 4 ? 3 - q : 4;
				}
				prof[j] = q > 4 ? 0u : ((uint32_t)(uint8_t)mat[q] | (uint32_t)(uint8_t)mat[5 + q] << 8 | (uint32_t)(uint8_t)mat[10 + q] << 16 | (uint32_t)(uint8_t)mat[15 + q] << 24);
			}
			int gmax = 0, te = -1, n_b = 0;
			unsigned long long b_last = 0;
			unsigned int tw = 0;     // the target bases of 16 rows, two bits each, the same in every lane of the row
			bool live = run && tlen > 0;
			int rows = live ? tlen : 0;
			rows = wave_max_i32(rows);   // the wave walks as many rows as its longest job has
			for (int i = 0; i < rows; ++i) {
				if (!__builtin_amdgcn_readfirstlane(__ballot(live) != 0)) break;
				if (live && i >= tlen) live = false;
				if (live) {
					if ((i & 15) == 0) {
						const int ii = i + k;
						const long long src = (ii <= te_rev) ? (long long)(te_rev - ii) : (long long)ii;
						const unsigned int tb = ii < tlen ? (unsigned int)dev_ref_base(ix.pac, ix.l_pac, J.tpos + src * J.tdir) : 0u;   // (the reference never holds an ambiguous base, bntseq.c:558-559)
						tw = swl_row_or(tb << (k << 1));
					}
					const int tsh = (int)((tw >> ((i & 15) << 1)) & 3u) << 3;
					// ---- the striped main loop and what lazy F converges to: with the same stripe count in every row of the wave (the launch's order puts like with
					// like; the second pass's counts differ) the stripe tests are scalar branches
					int mxv;
					if (uni) mxv = swl_row<SL, true>(H, E, prof, tsh, cap, slen_s, oe_del, e_del, oe_ins, e_ins, k * slen_s * e_ins);
					else mxv = swl_row<SL, false>(H, E, prof, tsh, cap, slen, oe_del, e_del, oe_ins, e_ins, k * slen * e_ins);
					// ---- the row's maximum, b[], the best row so far (ksw.c:188-207)
					const int imax = swl_row_max(mxv);
					if (imax >= minsc) {
						if (n_b == 0 || (int)(uint32_t)b_last + 1 != i) { b_last = (unsigned long long)imax << 32 | (uint32_t)i; ++n_b; }
						else if ((int)(b_last >> 32) < imax) b_last = (unsigned long long)imax << 32 | (uint32_t)i;
						if (k == 0) b[n_b - 1] = b_last;
					}
					if (imax > gmax) {
						gmax = imax; te = i;
#pragma unroll
						for (int j = 0; j < SL; ++j) Hm[j] = H[j];
						if (gmax + shift >= 255 || gmax >= endsc) live = false;
					}
				}
			}
			// ---- the pass's result (ksw.c:208-228)
			const int score = gmax + shift < 255 ? gmax : 255;
			int qe = -1, score2 = -1, te2 = -1;
			if (run && score != 255) {
				int lm = -1, lj = 0x7fffffff;
#pragma unroll
				for (int j = 0; j < SL; ++j) if (j < slen) { const int c = j + k * slen; if (Hm[j] > lm || (Hm[j] == lm && c < lj)) { lm = Hm[j]; lj = c; } }
				const int m = swl_row_max(lm);
				qe = swl_row_min(lm == m ? lj : 0x7fffffff);   // the smallest query column holding the maximum (ksw.c:212-216)
				if (n_b > 0) {
					WAVE_SYNC();
					const int rad = (score + mx - 1) / mx, low = te - rad, high = te + rad;
					int bs = -1, bi = 0x7fffffff;
					for (int x = k; x < n_b; x += 16) {
						const unsigned long long v = b[x];
						const int e = (int)(uint32_t)v, s2 = (int)(v >> 32);
						if ((e < low || e > high) && s2 > bs) { bs = s2; bi = x; }
					}
					const int ms = swl_row_max(bs);
					if (ms >= 0) {
						const int mi = swl_row_min(bs == ms ? bi : 0x7fffffff);   // the first entry with that score (ksw.c:222-225)
						score2 = ms; te2 = (int)(uint32_t)b[mi];
					}
					WAVE_SYNC();
				}
			}
			if (pass == 0) {
				o.score = score; o.te = te; o.qe = qe; o.score2 = score2; o.te2 = te2;
				r_score = score; r_te = te; r_qe = qe;
				run = run && (J.xtra & BSX_KSW_XSTART) && !((J.xtra & BSX_KSW_XSUBO) && score < (J.xtra & 0xffff)) && qe >= 0;
				if (__ballot(run) == 0) break;
			} else if (run && r_score == score) { o.tb = r_te - te; o.qb = r_qe - qe; }
		}
		if (valid && k == 0) res[job] = o;
	}
}

void launch_swl(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order,
                long long n, bsx_sw_res_t *res, unsigned long long *bscratch, int bcap, int blocks, int slen_max)
{
	if (slen_max <= 10) hipLaunchKernelGGL((k_swl<10>), dim3(blocks), dim3(256), 0, st, ix, sc, reads, jobs, order, n, res, bscratch, bcap);
	else hipLaunchKernelGGL((k_swl<16>), dim3(blocks), dim3(256), 0, st, ix, sc, reads, jobs, order, n, res, bscratch, bcap);
}
