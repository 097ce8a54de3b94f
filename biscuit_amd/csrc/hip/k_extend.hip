// k_extend.hip -- K4: banded affine-gap seed extension, ksw_extend2 (lib/aln/ksw.c:380-479).
//
// One wavefront per extension job; 64 lanes sweep the active band of one target row at a time.
// The reference opens gaps from the diagonal term M = H(i-1,j-1)+s and not from H(i,j)
// (ksw.c:433-447), so within a row
//     F(j) = max_{k<j} ( max(M(k)-oe_ins,0) - (j-1-k)*e_ins )
// is a pure max-plus prefix scan of previous-row data: a row needs one wave scan per 64 columns,
// rows stay sequential.  H/E live in LDS in exactly the reference's in-place eh[] layout
// (eh[j].h = H(i-1,j-1), eh[j].e = E(i,j)) because the adaptive band can re-grow over cells it
// skipped earlier and must then see what the reference sees there.  All reads of a row happen
// before any write of that row, which makes the parallel sweep equivalent to the sequential one.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

template <int NC>
__global__ void __launch_bounds__(256)
k_extend(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order, long long n,
         bsx_ext_res_t *res, int qcap)
{
	extern __shared__ int32_t lds[];
	const int lane = wave_lane();
	const int waves_per_block = blockDim.x >> 6;
	const int wave = threadIdx.x >> 6;
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1;
	int32_t *H = lds + wave * stride;
	int32_t *E = H + (qcap + 2);
	uint8_t *qb = reinterpret_cast<uint8_t*>(E + (qcap + 2));

	for (long long jj = (long long)blockIdx.x * waves_per_block + wave; jj < n; jj += (long long)gridDim.x * waves_per_block) {
		const int job = order[jj];
		const bsx_ext_job_t J = jobs[job];
		const int qlen = J.qlen, tlen = J.tlen, h0 = J.h0;
		const int8_t *mat = J.parent ? sc.ctmat : sc.gamat;
		const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
		const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = sc.zdrop;
		// query into LDS; first row (ksw.c:395-397): H[0]=h0, H[j]=max(h0-oe_ins-(j-1)e_ins,0)
		for (int j = lane; j < qlen; j += 64) qb[j] = reads[(long long)J.qoff + (long long)j * J.qdir];
		for (int j = lane; j <= qlen; j += 64) {
			int v = j == 0 ? h0 : h0 - oe_ins - (j - 1) * e_ins;
			H[j] = v > 0 ? v : 0; E[j] = 0;
		}
		// band clamp (ksw.c:399-407)
		int mx = 0;
		for (int k = 0; k < 25; ++k) mx = mx > mat[k] ? mx : mat[k];
		int w = J.w;
		{
			int max_ins = (int)((double)(qlen * mx + J.end_bonus - o_ins) / e_ins + 1.);
			max_ins = max_ins > 1 ? max_ins : 1;
			w = w < max_ins ? w : max_ins;
			int max_del = (int)((double)(qlen * mx + J.end_bonus - o_del) / e_del + 1.);
			max_del = max_del > 1 ? max_del : 1;
			w = w < max_del ? w : max_del;
		}
		int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
		int beg = 0, end = qlen;
		int tb_reg = 4;
		WAVE_SYNC();
		for (int i = 0; i < tlen; ++i) {
			if ((i & 63) == 0) tb_reg = (i + lane < tlen) ? dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)(i + lane) * J.tdir) : 4;
			const int t = __builtin_amdgcn_readfirstlane(__shfl(tb_reg, i & 63));
			const int8_t s0 = mat[t * 5], s1 = mat[t * 5 + 1], s2 = mat[t * 5 + 2], s3 = mat[t * 5 + 3], s4 = mat[t * 5 + 4];
			if (beg < i - w) beg = i - w;
			if (end > i + w + 1) end = i + w + 1;
			if (end > qlen) end = qlen;
			int h1_init = 0;
			if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
			int m = 0, mj = -1, h1_last = h1_init;
			if (beg < end) {
				const int nch = (end - beg + 63) >> 6;
				int M[NC], Ev[NC];
				// pass A: read the whole row's inputs
#pragma unroll
				for (int c = 0; c < NC; ++c) {
					M[c] = 0; Ev[c] = 0;
					if (c < nch) {
						const int j = beg + (c << 6) + lane;
						if (j < end) {
							const int hp = H[j], q = qb[j];
							const int s = q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : q == 3 ? s3 : s4;
							M[c] = hp ? hp + s : 0;
							Ev[c] = E[j];
						}
					}
				}
				WAVE_SYNC();
				// pass B: F by prefix scan, H, E', row maximum
				int carry = NEG_BIG, lm = -1, lj = -1;
#pragma unroll
				for (int c = 0; c < NC; ++c) {
					if (c < nch) {
						const int j = beg + (c << 6) + lane;
						const bool act = j < end;
						int tins = M[c] - oe_ins; tins = tins > 0 ? tins : 0;
						const int g = act ? tins + j * e_ins : NEG_BIG;
						const int incl = wave_scan_max_incl(g);
						int excl = wave_prev(incl, NEG_BIG);
						excl = excl > carry ? excl : carry;            // prefix max over all earlier columns
						{ const int tot = __shfl(incl, 63); carry = carry > tot ? carry : tot; }
						int f = j == beg ? 0 : excl - (j - 1) * e_ins;
						if (f < 0) f = 0;
						if (act) {
							int h = M[c] > Ev[c] ? M[c] : Ev[c];
							h = h > f ? h : f;
							int tdel = M[c] - oe_del; tdel = tdel > 0 ? tdel : 0;
							int e = Ev[c] - e_del; e = e > tdel ? e : tdel;
							E[j] = e;
							H[j + 1] = h;
							if (j == beg) H[beg] = h1_init;
							if (j == end - 1) E[end] = 0;
							if (h >= lm) { lm = h; lj = j; }
						}
					}
				}
				WAVE_SYNC();
				m = wave_max_i32(lm);
				mj = wave_max_i32(lm == m ? lj : -1);
				h1_last = H[end];
			} else { // empty row: only the boundary cell is written (ksw.c:449)
				if (lane == 0) { H[end] = h1_init; E[end] = 0; }
				WAVE_SYNC();
			}
			const int jfin = beg < end ? end : beg;
			if (jfin == qlen) { max_ie = gscore > h1_last ? max_ie : i; gscore = gscore > h1_last ? gscore : h1_last; }
			if (m == 0) break;
			if (m > max) {
				max = m; max_i = i; max_j = mj;
				int off = mj - i; off = off < 0 ? -off : off;
				max_off = max_off > off ? max_off : off;
			} else if (zdrop > 0) {
				if (i - max_i > mj - max_j) { if (max - m - ((i - max_i) - (mj - max_j)) * e_del > zdrop) break; }
				else { if (max - m - ((mj - max_j) - (i - max_i)) * e_ins > zdrop) break; }
			}
			// shrink the band to the non-zero cells (ksw.c:466-469), reading back what was written:
			// beg' = first j in [beg,end) with (H,E) != 0 (else end); end' = last such j in [beg',end] + 2
			{
				const int nchk = (end - beg + 1 + 63) >> 6;   // cells beg..end inclusive
				int nb = end, last;
				for (int c = 0; c < nchk; ++c) {
					const int j = beg + (c << 6) + lane;
					const bool nz = j < end && (H[j] != 0 || E[j] != 0);
					const unsigned long long b = __ballot(nz);
					if (b) { nb = beg + (c << 6) + __builtin_ctzll(b); break; }
				}
				last = nb - 1;
				for (int c = nchk - 1; c >= 0; --c) {
					const int j = beg + (c << 6) + lane;
					const bool nz = j <= end && j >= nb && (H[j] != 0 || E[j] != 0);
					const unsigned long long b = __ballot(nz);
					if (b) { last = beg + (c << 6) + 63 - __builtin_clzll(b); break; }
				}
				beg = nb;
				end = last + 2 < qlen ? last + 2 : qlen;
			}
			WAVE_SYNC();
		}
		if (lane == 0) {
			bsx_ext_res_t r;
			r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
			res[job] = r;
		}
		WAVE_SYNC();
	}
}

template <int NC>
static void launch_ext_nc(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs,
                          const int *order, long long n, bsx_ext_res_t *res, int qcap, int n_cu)
{
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1;
	int wpb = 4;
	while (wpb > 1 && (size_t)wpb * stride * 4 > 60 * 1024) wpb >>= 1;
	const size_t lds = (size_t)wpb * stride * 4;
	long long blocks = (n + wpb - 1) / wpb;
	long long cap = (long long)n_cu * 32;
	if (blocks > cap) blocks = cap;
	if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_extend<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k_extend<NC>, dim3((unsigned)blocks), dim3(wpb * 64), lds, st, ix, sc, reads, jobs, order, n, res, qcap);
}

void launch_extend(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order,
                   long long n, bsx_ext_res_t *res, int qcap, int nc, int n_cu)
{
	if (nc <= 4) launch_ext_nc<4>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
	else if (nc <= 8) launch_ext_nc<8>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
	else launch_ext_nc<32>(st, ix, sc, reads, jobs, order, n, res, qcap, n_cu);
}
