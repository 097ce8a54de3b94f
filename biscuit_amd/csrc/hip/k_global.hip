// k_global.hip -- K6: banded global alignment with traceback, ksw_global2 (lib/aln/ksw.c:504-606),
// wrapped in the band set-up of bis_bwa_gen_cigar2 (lib/aln/bwa.c:314-340) and the band-doubling
// retry loop of mem_alnreg_setSAM (lib/aln/mem_alnreg_format.c:63-77).
//
// One wavefront per job.  As in ksw_extend2, gaps open from M = H(i-1,j-1)+s, so F along a row is a
// max-plus prefix scan and the 64 lanes sweep the band [i-w, i+w] of one target row per step.
// H/E rows sit in LDS in the reference's in-place eh[] layout; the 1-byte/cell direction matrix
// z[tlen][n_col] is streamed to a per-wave slab in HBM (it does not fit LDS for long reads) and
// walked back by lane 0.
// The job's final CIGAR is then walked over the job's own sequences for the second half of bis_bwa_gen_cigar2
// (lib/aln/bwa.c:342-418): NM, the MD string, BISCUIT's conversion (ZC) and retention (ZR) counts.  The read is already in
// LDS; the target bases are staged there by the wave when they fit; lane 0 walks twice (lengths, then bytes: the strings
// are packed, each job taking exactly its length from one cursor).
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "wave.hpp"
#include "kernels.h"

#define G_MINUS_INF (-0x40000000)
#define G_INACTIVE  (-0x7f000000)

template <int NC>
__global__ void __launch_bounds__(256)
k_global(DevIndex ix, DevScoring sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order, long long n,
         bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap,
         bsx_glb_tag_t *tags, char *md_pool, unsigned long long md_cap, unsigned long long *md_cursor, int tcap, int32_t *hbm_rows)
{
	extern __shared__ int32_t lds[];
	const int lane = wave_lane();
	const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1 + ((tcap + 3) >> 2);
	// hbm_rows: queries too long for LDS keep their rows in a per-wave slab in HBM (as k_extend)
	int32_t *H = hbm_rows ? hbm_rows + ((size_t)blockIdx.x * wpb + wave) * (size_t)stride : lds + wave * stride;
	int32_t *E = H + (qcap + 2);
	uint8_t *qb = reinterpret_cast<uint8_t*>(E + (qcap + 2));
	uint8_t *tb = qb + (((qcap + 3) >> 2) << 2) + 4;   // tcap target bases (tags only)
	uint8_t *z = zscratch + ((size_t)blockIdx.x * wpb + wave) * zstride;

	for (long long jj = (long long)blockIdx.x * wpb + wave; jj < n; jj += (long long)gridDim.x * wpb) {
		const int job = order[jj];
		const bsx_glb_job_t J = jobs[job];
		const int qlen = J.qlen, tlen = J.tlen;
		const int8_t *mat = J.use_ct ? sc.ctmat : sc.gamat;
		const int o_del = sc.o_del, e_del = sc.e_del, o_ins = sc.o_ins, e_ins = sc.e_ins;
		const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
		uint32_t *cig = pool + J.cigar_off;
		for (int j = lane; j < qlen; j += 64) qb[j] = reads[(long long)J.qoff + (long long)j * J.qdir];
		WAVE_SYNC();
		int w_ = J.w0, score = 0, last_sc = -(1 << 30), n_cigar = 0, w_used = 0;
		for (int it = 0; it < J.n_try; ++it, w_ <<= 1, last_sc = score) {
			w_ = w_ < J.w_max ? w_ : J.w_max;
			w_used = w_;
			if (qlen == tlen && w_ == 0) { // no gap: one M run, score by direct comparison (bwa.c:314-322)
				int part = 0;
				for (int k = lane; k < qlen; k += 64) {
					const int t = dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)k * J.tdir);
					part += mat[t * 5 + qb[k]];
				}
				score = wave_sum_i32(part);
				if (J.want_cigar) { if (J.cigar_cap >= 1) { if (lane == 0) cig[0] = (uint32_t)qlen << 4; n_cigar = 1; } else n_cigar = -1; }
			} else {
				// band (bwa.c:326-335)
				int dl = tlen - qlen; dl = dl < 0 ? -dl : dl;
				int max_ins = (int)((double)(((qlen + 1) >> 1) * mat[0] - o_ins) / e_ins + 1.);
				int max_del = (int)((double)(((qlen + 1) >> 1) * mat[0] - o_del) / e_del + 1.);
				int max_gap = max_ins > max_del ? max_ins : max_del;
				max_gap = max_gap > 1 ? max_gap : 1;
				int w = (max_gap + dl + 1) >> 1;
				w = w < w_ ? w : w_;
				{ const int min_w = dl + 3; w = w > min_w ? w : min_w; }
				const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
				// first row (ksw.c:522-526)
				for (int j = lane; j <= qlen; j += 64) {
					if (j == 0) { H[0] = 0; E[0] = G_MINUS_INF; }
					else if (j <= w) { H[j] = -(o_ins + e_ins * j); E[j] = G_MINUS_INF; }
					else { H[j] = G_MINUS_INF; E[j] = G_MINUS_INF; }
				}
				WAVE_SYNC();
				int tb_reg = 4;
				for (int i = 0; i < tlen; ++i) {
					if ((i & 63) == 0) tb_reg = (i + lane < tlen) ? dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)(i + lane) * J.tdir) : 4;
					const int t = wave_bcast(tb_reg, i & 63);
					const int s0 = mat[t * 5], s1 = mat[t * 5 + 1], s2 = mat[t * 5 + 2], s3 = mat[t * 5 + 3], s4 = mat[t * 5 + 4];
					const int beg = i > w ? i - w : 0;
					const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
					const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : G_MINUS_INF;
					if (beg < end) {
						const int nch = (end - beg + 63) >> 6;
						int M[NC], Ev[NC];
#pragma unroll
						for (int c = 0; c < NC; ++c) {
							M[c] = G_MINUS_INF; Ev[c] = G_MINUS_INF;
							if (c < nch) {
								const int j = beg + (c << 6) + lane;
								if (j < end) {
									const int q = qb[j];
									const int s = q == 0 ? s0 : q == 1 ? s1 : q == 2 ? s2 : q == 3 ? s3 : s4;
									M[c] = H[j] + s; Ev[c] = E[j];
								}
							}
						}
						WAVE_SYNC();
						int carry = G_INACTIVE;
						uint8_t *zi = z + (size_t)i * n_col;
#pragma unroll
						for (int c = 0; c < NC; ++c) {
							if (c < nch) {
								const int j = beg + (c << 6) + lane;
								const bool act = j < end;
								const int g = act ? M[c] - oe_ins + j * e_ins : G_INACTIVE;
								const int incl = wave_scan_max_incl(g);
								int excl = wave_prev(incl, G_INACTIVE);
								excl = excl > carry ? excl : carry;
								{ const int tot = __builtin_amdgcn_readlane(incl, 63); carry = carry > tot ? carry : tot; }
								const int f = j == beg ? G_MINUS_INF : excl - (j - 1) * e_ins;
								if (act) {
									const int m = M[c], e0 = Ev[c];
									int d = m >= e0 ? 0 : 1;
									int h = m >= e0 ? m : e0;
									d = h >= f ? d : 2;
									h = h >= f ? h : f;
									int tt = m - oe_del, e = e0 - e_del;
									d |= e > tt ? 1 << 2 : 0;
									e = e > tt ? e : tt;
									tt = m - oe_ins;
									d |= (f - e_ins) > tt ? 2 << 4 : 0;
									E[j] = e;
									H[j + 1] = h;
									if (j == beg) H[beg] = h1_init;
									if (j == end - 1) E[end] = G_MINUS_INF;
									if (J.want_cigar) zi[j - beg] = (uint8_t)d;
								}
							}
						}
						WAVE_SYNC();
					} else {
						if (lane == 0) { H[end] = h1_init; E[end] = G_MINUS_INF; }
						WAVE_SYNC();
					}
				}
				score = H[qlen];
				if (J.want_cigar) { // traceback (ksw.c:587-604), lane 0
					__threadfence_block();
					WAVE_SYNC();
					int n = 0;
					if (lane == 0) {
						int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
						uint32_t cur = 0; // pending run: len<<4|op
						const int cap = (int)J.cigar_cap;
						#define GPUSH(op, len) do { if (cur != 0 && (cur & 0xf) == (uint32_t)(op)) cur += (uint32_t)(len) << 4; \
							else { if (cur != 0) { if (n < cap) cig[n] = cur; ++n; } cur = (uint32_t)(len) << 4 | (uint32_t)(op); } } while (0)
						while (i >= 0 && k >= 0) {
							const int d = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))];
							which = d >> (which << 1) & 3;
							if (which == 0) { GPUSH(0, 1); --i; --k; }
							else if (which == 1) { GPUSH(2, 1); --i; }
							else { GPUSH(1, 1); --k; }
						}
						if (i >= 0) GPUSH(2, i + 1);
						if (k >= 0) GPUSH(1, k + 1);
						if (cur != 0) { if (n < cap) cig[n] = cur; ++n; }
						#undef GPUSH
						if (n <= cap) for (int a = 0; a < n >> 1; ++a) { const uint32_t tmp = cig[a]; cig[a] = cig[n - 1 - a]; cig[n - 1 - a] = tmp; }
						else n = -n;
					}
					n_cigar = __builtin_amdgcn_readlane(n, 0);
				}
			}
			if (J.n_try == 1) break;
			if (score == last_sc) break;
			if (w_ == J.w_max) break;
			if (score >= J.truesc - sc.a) break;
		}
		if (lane == 0) { bsx_glb_res_t r; r.score = score; r.n_cigar = n_cigar; r.w_used = w_used; r.pad = 0; res[job] = r; }
		if (tags && J.want_cigar) { // MD / NM / ZC / ZR (bwa.c:342-418)
			const bool staged = tlen <= tcap;
			const int parent = J.use_ct, rev = J.tdir < 0;
			#define MD_BASE(r) ((r) > 3 ? 'N' : (int)(((rev ? 0x41434754u : 0x54474341u) >> ((r) << 3)) & 0xffu))   /* "ACGT" / "TGCA" (bwa.c:297,345) */
			#define MD_DIGITS(v) ((v) >= 10000 ? 5 : (v) >= 1000 ? 4 : (v) >= 100 ? 3 : (v) >= 10 ? 2 : 1)
			__threadfence_block();   // the CIGAR lane 0 wrote, read back below
			WAVE_SYNC();
			if (n_cigar > 0 && staged) {
				// The wave walks the alignment 64 columns at a time: a lane compares one column; a mismatching lane knows the run of
				// matches before it from the ballot (the lanes below it, or the run carried in from the columns before), hence the
				// length of its "<run><base>" and, by a prefix sum, where it goes.  The string is assembled in the LDS the DP rows
				// occupied (8 bytes per query column >= 2 per target base) and copied out once its length is known.
				for (int k = lane; k < tlen; k += 64) tb[k] = (uint8_t)dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)k * J.tdir);
				uint8_t *mdb = reinterpret_cast<uint8_t*>(H);
				WAVE_SYNC();
				int x = 0, y = 0, u = 0, l = 0, n_mm = 0, n_gap = 0, n_conv = 0, n_ret = 0;
				for (int k = 0; k < n_cigar; ++k) {
					const uint32_t cg = (uint32_t)__builtin_amdgcn_readfirstlane((int)cig[k]);
					const int op = (int)(cg & 0xf), len = (int)(cg >> 4);
					if (op == 0) {
						for (int b0 = 0; b0 < len; b0 += 64) {
							const int n = len - b0 < 64 ? len - b0 : 64;
							const bool act = lane < n;
							const int q = act ? qb[x + b0 + lane] : 0, r = act ? tb[y + b0 + lane] : 0;
							const bool mm = act && q != r;
							const bool conv = mm && (parent ? (q == 3 && r == 1) : (q == 0 && r == 2));
							const unsigned long long B = __ballot(mm), Bc = __ballot(conv);
							n_ret += __popcll(__ballot(act && q == r && (parent ? q == 1 : q == 2)));
							n_conv += __popcll(Bc); n_mm += __popcll(B) - __popcll(Bc);
							if (B) {
								const unsigned long long below = B & ((1ull << lane) - 1ull);
								const int prev = below ? 63 - __builtin_clzll(below) : -1;
								int run = lane - prev - 1 + (prev < 0 ? u : 0);
								const int nd = MD_DIGITS(run), mine = mm ? nd + 1 : 0;
								const int incl = wave_scan_sum_incl(mine);
								if (mm) {
									uint8_t *d = mdb + l + incl - mine;
									d[nd] = (uint8_t)MD_BASE(r);
									for (int t = nd - 1; t >= 0; --t) { d[t] = (uint8_t)('0' + run % 10); run /= 10; }
								}
								l += __builtin_amdgcn_readlane(incl, 63);
								u = n - 1 - (63 - __builtin_clzll(B));
							} else u += n;
						}
						x += len; y += len;
					} else if (op == 2) {
						if (k > 0 && k < n_cigar - 1) {
							const int nd = MD_DIGITS(u);
							if (lane == 0) { int v = u; mdb[l + nd] = '^'; for (int t = nd - 1; t >= 0; --t) { mdb[l + t] = (uint8_t)('0' + v % 10); v /= 10; } }
							l += nd + 1;
							for (int i = lane; i < len; i += 64) { const int r = tb[y + i]; mdb[l + i] = (uint8_t)MD_BASE(r); }
							l += len; u = 0; n_gap += len;
						}
						y += len;
					} else if (op == 1) { x += len; n_gap += len; }
				}
				{
					const int nd = MD_DIGITS(u);
					if (lane == 0) { int v = u; mdb[l + nd] = 0; for (int t = nd - 1; t >= 0; --t) { mdb[l + t] = (uint8_t)('0' + v % 10); v /= 10; } }
					l += nd;
				}
				WAVE_SYNC();
				unsigned long long at = 0;
				if (lane == 0) at = atomicAdd(md_cursor, (unsigned long long)l + 1);
				at = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(at >> 32), 0) << 32 | (unsigned)__builtin_amdgcn_readlane((int)at, 0);
				if (at + (unsigned long long)l + 1 <= md_cap) {   // always: the pool holds every job's upper bound
					for (int i = lane; i <= l; i += 64) md_pool[at + i] = (char)mdb[i];
					if (lane == 0) {
						bsx_glb_tag_t T; T.NM = n_mm + n_gap; T.ZC = n_conv; T.ZR = n_ret; T.l_md = l; T.md_off = at; T.bss_u = n_conv == 0;
						for (int k = 0; k < 7; ++k) T.pad[k] = 0;
						tags[job] = T;
					}
				}
			} else
			if (lane == 0) { // no CIGAR (l_md = -1), or a target longer than the LDS stage: one lane, twice (lengths, then bytes)
				bsx_glb_tag_t T; T.NM = T.ZC = T.ZR = 0; T.l_md = -1; T.md_off = 0; T.bss_u = 0;
				for (int k = 0; k < 7; ++k) T.pad[k] = 0;
				if (n_cigar > 0) {
					unsigned long long at = 0;
					char *dst = nullptr;
					for (int pass = 0; pass < 2; ++pass) {
						int x = 0, y = 0, u = 0, l = 0, n_mm = 0, n_gap = 0, n_conv = 0, n_ret = 0;
						// decimal of the pending run of matches, then one character
						#define MD_NUM(v) do { int v_ = (v), nd_ = v_ >= 10000 ? 5 : v_ >= 1000 ? 4 : v_ >= 100 ? 3 : v_ >= 10 ? 2 : 1; \
							if (dst) { int t_ = v_; for (int d_ = nd_ - 1; d_ >= 0; --d_) { dst[l + d_] = (char)('0' + t_ % 10); t_ /= 10; } } l += nd_; } while (0)
						#define MD_CHR(c) do { if (dst) dst[l] = (char)(c); ++l; } while (0)
						for (int k = 0; k < n_cigar; ++k) {
							const int op = (int)(cig[k] & 0xf), len = (int)(cig[k] >> 4);
							if (op == 0) {
								for (int i = 0; i < len; ++i) {
									const int q = qb[x + i];
									const int r = dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)(y + i) * J.tdir);
									if (q == r) { n_ret += parent ? q == 1 : q == 2; ++u; }
									else {
										MD_NUM(u); MD_CHR(MD_BASE(r)); u = 0;
										if (parent ? (q == 3 && r == 1) : (q == 0 && r == 2)) ++n_conv; else ++n_mm;
									}
								}
								x += len; y += len;
							} else if (op == 2) {
								if (k > 0 && k < n_cigar - 1) {
									MD_NUM(u); MD_CHR('^');
									for (int i = 0; i < len; ++i) { const int r = dev_ref_base(ix.pac, ix.l_pac, J.tpos + (long long)(y + i) * J.tdir); MD_CHR(MD_BASE(r)); }
									u = 0; n_gap += len;
								}
								y += len;
							} else if (op == 1) { x += len; n_gap += len; }
						}
						MD_NUM(u);
						#undef MD_NUM
						#undef MD_CHR
						if (pass == 0) {
							at = atomicAdd(md_cursor, (unsigned long long)l + 1);
							if (at + (unsigned long long)l + 1 > md_cap) break;   // cannot happen: the pool holds every job's upper bound
							dst = md_pool + at;
							T.NM = n_mm + n_gap; T.ZC = n_conv; T.ZR = n_ret; T.bss_u = n_conv == 0; T.l_md = l; T.md_off = at;
						} else dst[l] = 0;
					}
				}
				tags[job] = T;
			}
			#undef MD_BASE
			#undef MD_DIGITS
		}
		WAVE_SYNC();
	}
}

template <int NC>
static void launch_glb_nc(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order,
                          long long n, bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap, int blocks, int wpb,
                          bsx_glb_tag_t *tags, char *md_pool, unsigned long long md_cap, unsigned long long *md_cursor, int tcap, void *hbm_rows = nullptr)
{
	const int stride = 2 * (qcap + 2) + ((qcap + 3) >> 2) + 1 + ((tcap + 3) >> 2);
	const size_t lds = hbm_rows ? 0 : (size_t)wpb * stride * 4;
	if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_global<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	hipLaunchKernelGGL(k_global<NC>, dim3(blocks), dim3(wpb * 64), lds, st, ix, sc, reads, jobs, order, n, res, pool, zscratch, zstride, qcap,
	                   tags, md_pool, md_cap, md_cursor, tcap, (int32_t*)hbm_rows);
}
size_t global_hbm_row_bytes(int qcap) { return (size_t)(2 * (qcap + 2) + ((qcap + 3) >> 2) + 1) * 4; }
void launch_global_hbm(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order,
                       long long n, bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap, int blocks,
                       bsx_glb_tag_t *tags, char *md_pool, unsigned long long md_cap, unsigned long long *md_cursor, void *rows)
{
	launch_glb_nc<32>(st, ix, sc, reads, jobs, order, n, res, pool, zscratch, zstride, qcap, blocks, 1, tags, md_pool, md_cap, md_cursor, 0, rows);
}

void launch_global(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order,
                   long long n, bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap, int nc, int blocks, int wpb,
                   bsx_glb_tag_t *tags, char *md_pool, unsigned long long md_cap, unsigned long long *md_cursor, int tcap)
{
	if (nc <= 4) launch_glb_nc<4>(st, ix, sc, reads, jobs, order, n, res, pool, zscratch, zstride, qcap, blocks, wpb, tags, md_pool, md_cap, md_cursor, tcap);
	else if (nc <= 16) launch_glb_nc<16>(st, ix, sc, reads, jobs, order, n, res, pool, zscratch, zstride, qcap, blocks, wpb, tags, md_pool, md_cap, md_cursor, tcap);
	else launch_glb_nc<32>(st, ix, sc, reads, jobs, order, n, res, pool, zscratch, zstride, qcap, blocks, wpb, tags, md_pool, md_cap, md_cursor, tcap);
}
