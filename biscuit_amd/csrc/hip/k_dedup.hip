// k_dedup.hip -- C5 on the device: mem_sort_deduplicate (lib/aln/mem_alnreg.c:112-202) over the regions the region tiers
// left in HBM, one LANE per read.
//
// The work of a read is a few dozen dependent steps over a handful of regions (two sorts with klib's introsort, whose
// order for equal keys matters; a backward scan per region with early exits): nothing in it spans lanes, and a chunk has a
// million reads, so each lane runs the reference's own loop nest for one read and the parallelism is the number of reads.
// The lane keeps its order array in LDS and reads the regions' fields where they lie (no per-lane copies: those were 528
// bytes of scratch memory and 256 registers).
// A read's regions are the regions of its strand searches concatenated in call order (bwamem.c:352-372); without a
// concatenation (below) the function only drops and reorders them, so the result is a list of indices into that
// concatenation: the host builds the read's mem_alnreg_v straight in its final order.
// Left to the host (out_n = -1): a read one of whose strand searches the device did not finish (declined, or being seeded
// again), more than DD_CAP regions or DD_PER_READ strand searches, and a read where mem_test_reg_concatenation (mem_alnreg.c:63-108) gets as far as its
// global alignment -- two collinear regions a gap apart: the score is a k_global job, and the host's merge rounds batch those.
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "kernels.h"

#define DD_CAP 32

// klib introsort (ksort.h:184-236) over region indices with the control flow of csrc/host/util.c:bsx_introsort (see
// rg_introsort_keys in k_regions.hip); LT(x, y) compares the regions the indices name
template <typename Arr, typename LT>
__device__ __forceinline__ void dd_introsort(Arr a, int n, LT lt)
{
#define SWP(i, j) do { const unsigned char t_ = a[i]; a[i] = a[j]; a[j] = t_; } while (0)
	if (n < 2) return;
	if (n == 2) { if (lt(a[1], a[0])) SWP(0, 1); return; }
	int d, s = 0, t = n - 1, i, j, k, top = 0;
	int stk_l[4], stk_r[4], stk_d[4];   // a side is stacked only when longer than 16: at most one of a range of <= DD_CAP
	for (d = 2; (1 << d) < n; ++d);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { // comb sort fallback (ksort.h:162-183); not reached for n <= DD_CAP, kept for the control flow
				const double shrink = 1.2473309501039786540366528676643;
				int m = t - s + 1, gap = m, swapped;
				do {
					if (gap > 2) { gap = (int)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
					swapped = 0;
					for (i = 0; i + gap < m; ++i) if (lt(a[s + i + gap], a[s + i])) { SWP(s + i, s + i + gap); swapped = 1; }
				} while (swapped || gap > 2);
				if (gap != 1) for (i = s + 1; i <= t; ++i) for (j = i; j > s && lt(a[j], a[j - 1]); --j) SWP(j, j - 1);
				t = s;
				continue;
			}
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
			else k = lt(a[j], a[i]) ? i : j;
			const unsigned char rp = a[k];
			if (k != t) SWP(k, t);
			for (;;) {
				do ++i; while (lt(a[i], rp));
				do --j; while (i <= j && lt(rp, a[j]));
				if (j <= i) break;
				SWP(i, j);
			}
			SWP(i, t);
			if (i - s > t - i) {
				if (i - s > 16) { stk_l[top] = s; stk_r[top] = i - 1; stk_d[top] = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stk_l[top] = i + 1; stk_r[top] = t; stk_d[top] = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) { for (i = 1; i < n; ++i) for (j = i; j > 0 && lt(a[j], a[j - 1]); --j) SWP(j, j - 1); return; }
			--top; s = stk_l[top]; t = stk_r[top]; d = stk_d[top];
		}
	}
#undef SWP
}

// a read's regions, numbered through the concatenation of its strand searches' lists; the records stay where the region
// tiers left them (32 of their 56 bytes are read here, a few times each and out of L2), so that a lane carries a handful
// of registers and an order array of DD_CAP bytes in LDS instead of six arrays of DD_CAP entries in scratch memory
#define DD_PER_READ 4
struct DdRead {
	const bsx_region_t *base[DD_PER_READ]; int cum[DD_PER_READ];   // list t holds entries cum[t-1] .. cum[t]-1
	__device__ __forceinline__ const bsx_region_t &at(int k) const
	{
		const bsx_region_t *r = base[0] + k;
#pragma unroll
		for (int t = 1; t < DD_PER_READ; ++t) if (k >= cum[t - 1]) r = base[t] + (k - cum[t - 1]);
		return *r;
	}
};
struct DdOrd {   // the order array of one lane: entry i at p[i * 256]
	unsigned char *p;
	__device__ __forceinline__ unsigned char &operator[](int i) const { return p[i * 256]; }
};

__global__ void __launch_bounds__(256)
k_dedup(const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
        long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, int *out_n, unsigned char *out_idx)
{
	__shared__ unsigned char s_ord[DD_CAP * 256];
	const int rd = (int)(blockIdx.x * blockDim.x + threadIdx.x);
	if (rd >= n_reads) return;
	if (per_read > DD_PER_READ) { out_n[rd] = -1; return; }
	DdRead R;
	int n = 0;
#pragma unroll
	for (int t = 0; t < DD_PER_READ; ++t) {
		R.base[t] = regs; R.cum[t] = n;
		if (t < per_read) {
			const int m = reg_n[rd * per_read + t];
			if (m < 0 || n + m > DD_CAP) { out_n[rd] = -1; return; }
			R.base[t] = regs + reg_off[rd * per_read + t];
			n += m; R.cum[t] = n;
		}
	}
	unsigned char *o = out_idx + (size_t)rd * DD_CAP;
	if (n <= 1) { if (n) o[0] = 0; out_n[rd] = n; return; }   // mem_alnreg.c:114
	DdOrd ord; ord.p = s_ord + threadIdx.x;
	for (int k = 0; k < n; ++k) ord[k] = (unsigned char)k;
	dd_introsort(ord, n, [&](int x, int y) { return R.at(x).re < R.at(y).re; });   // by END (alnreg_slt2)
	unsigned int dead = 0;   // stands for qe = qb (mem_alnreg.c:137,139)
	for (int a = 1; a < n; ++a) {
		const int p = ord[a];
		const bsx_region_t &P = R.at(p);
		const long long rb_p = P.rb, re_p = P.re; const int qb_p = P.qb, qe_p = P.qe, rid_p = P.rid, sc_p = P.score;
		for (int b = a - 1; b >= 0; --b) {
			const int q = ord[b];
			const bsx_region_t &Q = R.at(q);
			const long long rb_q = Q.rb, re_q = Q.re; const int qb_q = Q.qb, qe_q = Q.qe;
			if (!(rid_p == Q.rid && rb_p < re_q + max_chain_gap)) break;
			if (dead >> q & 1) continue;
			const long long or_ = re_q - rb_p;
			const long long oq = qb_q < qb_p ? qe_q - qb_p : qe_p - qb_q;
			const long long mr = re_q - rb_q < re_p - rb_p ? re_q - rb_q : re_p - rb_p;
			const long long mq = qe_q - qb_q < qe_p - qb_p ? qe_q - qb_q : qe_p - qb_p;
			if ((float)or_ > mask_level_redun * (float)mr && (float)oq > mask_level_redun * (float)mq) { // one of the two is redundant
				if (sc_p < Q.score) { dead |= 1u << p; break; }
				else dead |= 1u << q;
			} else if (rb_q < rb_p) { // mem_test_reg_concatenation(q, p) up to its alignment (mem_alnreg.c:63-91)
				if (rb_q < l_pac && rb_p >= l_pac) continue;
				if (qb_q >= qb_p || qe_q >= qe_p || re_q >= re_p) continue;
				long long w = (re_q - rb_p) - (long long)(qe_q - qb_p);
				int wi = (int)w; wi = wi > 0 ? wi : -wi;
				double r = (double)(re_q - rb_p) / (double)(re_p - rb_q) - (double)(qe_q - qb_p) / (double)(qe_p - qb_q);
				r = r > 0. ? r : -r;
				if (re_q < rb_p || qe_q < qb_p) { if (wi > opt_w << 1 || r >= (double)0.05f) continue; }
				else if (wi > opt_w << 2 || r >= (double)(0.05f * 2)) continue;
				out_n[rd] = -1;   // the two may be one alignment: scored and merged by the host's rounds
				return;
			}
		}
	}
	int m = 0;
	for (int k = 0; k < n; ++k) { const unsigned char v = ord[k]; if (!(dead >> v & 1)) ord[m++] = v; }
	dd_introsort(ord, m, [&](int x, int y) {   // alnreg_slt
		const bsx_region_t &X = R.at(x), &Y = R.at(y);
		return X.score > Y.score || (X.score == Y.score && (X.rb < Y.rb || (X.rb == Y.rb && X.qb < Y.qb)));
	});
	dead = 0;
	for (int k = 1; k < m; ++k) { // identical hits
		const int x = ord[k]; const bsx_region_t &X = R.at(x), &Y = R.at(ord[k - 1]);
		if (X.score == Y.score && X.rb == Y.rb && X.qb == Y.qb) dead |= 1u << x;
	}
	int m2 = 0;
	for (int k = 0; k < m; ++k) { const unsigned char v = ord[k]; if (k == 0 || !(dead >> v & 1)) o[m2++] = v; }
	out_n[rd] = m2;
}

int dedup_cap(void) { return DD_CAP; }
void launch_dedup(hipStream_t st, const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
                  long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, int *out_n, unsigned char *out_idx)
{
	hipLaunchKernelGGL(k_dedup, dim3((n_reads + 255) / 256), dim3(256), 0, st, regs, reg_off, reg_n, n_reads, per_read, l_pac, max_chain_gap, opt_w, mask_level_redun, out_n, out_idx);
}
