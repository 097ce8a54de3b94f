// k_index.hip -- the FM index of a converted text built on the device: suffix array by prefix doubling over radix
// sorts, then the BWT with its occurrence blocks and the suffix-array samples, all without leaving HBM.
//
// Replaces, for texts of any length, what the reference does on the host in bwt_bwtgen / is_bwt + bwt_bwtupdate_core +
// bwt_cal_sa (lib/aln/bwtindex.c:206-347, bwt_gen.c:1595-1607, is.c:208-223, bwt.c:63-85).  BWT and suffix array of a
// text are unique, so the output is byte-identical to the reference's files whatever the sorter.  An hg38-sized index
// is two texts of 6.2 G symbols: a 64-bit problem, sized for the memory one MI355X has (288 GB):
//
//   T    the converted text [fwd ; revcomp(fwd)], 2 bits per base, 32 bases per u64 (first base in the top bits), so
//        that the 32-mer at any position is two loads and a funnel shift and compares as an integer          n/4 B
//   SA   suffix order being refined in place                                                               8 n B
//   ISA  rank of every suffix at the current depth h (the slot of its group's head)                         8 n B
//   U    the suffixes whose group still has more than one member: (group rank, slot, suffix), in slot order
//
//   round 0   suffixes are dealt into 4^7 buckets by their first 7 bases (histogram), consecutive buckets form batches of
//             <= 256 M suffixes; a batch is collected (32-mer key, suffix), radix-sorted on the key, written to its slots;
//             equal keys form groups; singletons are final.  For a genome-like text almost everything ends here.
//   round h   (h = 32, 64, 128, ...) every member i of U gets the key ISA[i + h]; U is sorted by (group, key) -- one
//             radix sort on 128-bit keys, in slices cut at group boundaries --, groups split where keys differ,
//             singletons leave U.  ceil(log2(longest repeat / 32)) rounds; each touches only what is still tied.
//   Suffixes running past the end of the text compare as if followed by a unique smallest sentinel: their key is what is
//   left of them (shorter = smaller), every real key is ranked above those.
//   emit      BWT symbol of row r = T[SA[r] - 1]; one thread per 128-symbol block gathers, counts and writes the block
//             in the file layout (4 x u64 running counts + 8 x u32 symbols, lib/aln/bwt.h:93-101); the running counts
//             come from one scan over the per-block counts; SA samples are strided copies.
//
// Sorting and scanning use rocPRIM's device primitives (plain library sorts, as rocBLAS would be for a plain GEMM);
// everything specific to the index is written here.  Ranks and positions are 64-bit throughout.
#include <hip/hip_runtime.h>
#include "tune.h"
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <algorithm>
#include <vector>
#include <time.h>
#include "devbuf.hpp"
#include "dev_common.hpp"
#include "index_build.h"

typedef unsigned long long u64;
typedef __uint128_t u128;

#define IX_BUCKET_BASES 7
#define IX_BUCKETS (1 << (2 * IX_BUCKET_BASES))
#define IX_KEY2_BITS 36          // ISA[i + h] + h + 1 < 2^36 for every text that fits the machine

#define IXCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[bsx-index] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return BSX_E_NODEVICE; } } while (0)
#define RCCHK(x) do { int rc_ = (x); if (rc_ != BSX_OK) return rc_; } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// text
// ---------------------------------------------------------------------------------------------------------------------
// T[w] = bases 32w .. 32w+31 of the converted text; words past the end are zero.  cnt[c] += occurrences of symbol c.
__global__ void __launch_bounds__(256)
k_ix_text(const uint8_t *pac, long long l_pac, int parent, u64 *T, u64 n_words, u64 *cnt)
{
	const u64 n = (u64)l_pac << 1;
	unsigned int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (u64)gridDim.x * blockDim.x) {
		u64 v = 0;
		for (int j = 0; j < 32; ++j) {
			const u64 i = (w << 5) + j;
			if (i >= n) break;
			int b = dev_ref_base(pac, l_pac, (long long)i);
			b = parent ? (b == 1 ? 3 : b) : (b == 2 ? 0 : b);   // C>T resp. G>A of both halves (lib/aln/bntseq.c:585-600)
			v |= (u64)b << (62 - 2 * j);
			c0 += b == 0; c1 += b == 1; c2 += b == 2; c3 += b == 3;
		}
		T[w] = v;
	}
	for (int off = 32; off > 0; off >>= 1) { c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); c2 += __shfl_down(c2, off); c3 += __shfl_down(c3, off); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&cnt[0], (u64)c0); atomicAdd(&cnt[1], (u64)c1); atomicAdd(&cnt[2], (u64)c2); atomicAdd(&cnt[3], (u64)c3); }
}

// the 32 bases from position i on (zero-padded past the end)
__device__ __forceinline__ u64 ix_key(const u64 *T, u64 i)
{
	const u64 w = i >> 5; const int s = (int)(i & 31) << 1;
	const u64 a = T[w];
	return s ? (a << s) | (T[w + 1] >> (64 - s)) : a;
}
__device__ __forceinline__ int ix_base(const u64 *T, u64 i) { return (int)(T[i >> 5] >> (62 - ((i & 31) << 1))) & 3; }

// ---------------------------------------------------------------------------------------------------------------------
// round 0: buckets, batches, 32-mer sort
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_ix_hist(const u64 *T, u64 n, u64 *hist)
{
	__shared__ unsigned int h[IX_BUCKETS];
	for (int k = threadIdx.x; k < IX_BUCKETS; k += 256) h[k] = 0;
	__syncthreads();
	const u64 n_words = (n + 31) >> 5;
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (u64)gridDim.x * blockDim.x) {
		const u64 a = T[w], b = T[w + 1];
		for (int j = 0; j < 32; ++j) {
			if ((w << 5) + j >= n) break;
			const u64 key = j ? (a << (2 * j)) | (b >> (64 - 2 * j)) : a;
			atomicAdd(&h[key >> (64 - 2 * IX_BUCKET_BASES)], 1u);
		}
	}
	__syncthreads();
	for (int k = threadIdx.x; k < IX_BUCKETS; k += 256) if (h[k]) atomicAdd(&hist[k], (u64)h[k]);
}

// every suffix whose bucket lies in [b_lo, b_hi): (32-mer, position) appended at *cursor (any order: the sort follows)
__global__ void __launch_bounds__(256)
k_ix_collect(const u64 *T, u64 n, unsigned int b_lo, unsigned int b_hi, u64 *keys, u64 *vals, u64 *cursor)
{
	__shared__ unsigned int s_off[256];
	__shared__ u64 s_base;
	const u64 n_words = (n + 31) >> 5;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	const u64 trips = (n_words + stride - 1) / stride;
	for (u64 t = 0; t < trips; ++t) {
		const u64 w = t * stride + (u64)blockIdx.x * blockDim.x + threadIdx.x;
		u64 a = 0, b = 0;
		unsigned int mask = 0;
		if (w < n_words) {
			a = T[w]; b = T[w + 1];
			for (int j = 0; j < 32; ++j) {
				if ((w << 5) + j >= n) break;
				const u64 key = j ? (a << (2 * j)) | (b >> (64 - 2 * j)) : a;
				const unsigned int bk = (unsigned int)(key >> (64 - 2 * IX_BUCKET_BASES));
				if (bk >= b_lo && bk < b_hi) mask |= 1u << j;
			}
		}
		const unsigned int mine = __popc(mask);
		// exclusive prefix of `mine` over the workgroup
		s_off[threadIdx.x] = mine;
		__syncthreads();
		for (int off = 1; off < 256; off <<= 1) {
			unsigned int v = threadIdx.x >= (unsigned)off ? s_off[threadIdx.x - off] : 0;
			__syncthreads();
			s_off[threadIdx.x] += v;
			__syncthreads();
		}
		if (threadIdx.x == 255) s_base = s_off[255] ? atomicAdd(cursor, (u64)s_off[255]) : 0;
		__syncthreads();
		u64 at = s_base + s_off[threadIdx.x] - mine;
		while (mask) {
			const int j = __ffs(mask) - 1;
			mask &= mask - 1;
			keys[at] = j ? (a << (2 * j)) | (b >> (64 - 2 * j)) : a;
			vals[at] = (w << 5) + j;
			++at;
		}
		__syncthreads();
	}
}

// head[u] = u if element u opens a group (its key differs from its predecessor's), else 0: the running maximum of it is
// the index of the group head
__global__ void __launch_bounds__(256)
k_ix_heads64(const u64 *keys, u64 m, u64 *head)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x)
		head[u] = (u == 0 || keys[u] != keys[u - 1]) ? u : 0;
}

// A sorted batch takes the slots [base, base + m): SA and ISA are written, and what is still tied is flagged.
__global__ void __launch_bounds__(256)
k_ix_place0(const u64 *keys, const u64 *vals, const u64 *first, u64 m, u64 base, u64 *SA, u64 *ISA, u64 *tied)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x) {
		const u64 i = vals[u];
		SA[base + u] = i;
		ISA[i] = base + first[u];
		const bool head = first[u] == u, next_head = u + 1 == m || keys[u + 1] != keys[u];
		tied[u] = (head && next_head) ? 0 : 1;
	}
}
// tied elements move to the end of U, in order: (group rank, slot, suffix)
__global__ void __launch_bounds__(256)
k_ix_compact0(const u64 *vals, const u64 *first, const u64 *tied, const u64 *where, u64 m, u64 base, u64 u_at, u64 *Ugrp, u64 *Uslot, u64 *Uidx)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x) {
		if (!tied[u]) continue;
		const u64 o = u_at + where[u];
		Ugrp[o] = base + first[u]; Uslot[o] = base + u; Uidx[o] = vals[u];
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// round h: refine the groups that are still tied
// ---------------------------------------------------------------------------------------------------------------------
// key of member i: the rank of the suffix h further on (what is left of the text if that lies past its end)
__global__ void __launch_bounds__(256)
k_ix_key2(const u64 *Ugrp, const u64 *Uidx, u64 m, const u64 *ISA, u64 n, u64 h, u128 *comp, u64 *vals)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x) {
		const u64 i = Uidx[u];
		const u64 k2 = i + h < n ? ISA[i + h] + h + 1 : n - i;
		comp[u] = (u128)Ugrp[u] << IX_KEY2_BITS | (u128)k2;
		vals[u] = i;
	}
}
// after the sort the element at position u takes the u-th slot of the slice (slots and groups did not move, only the
// suffixes inside each group did): head[u] = its slot if it opens a (new, finer) group
__global__ void __launch_bounds__(256)
k_ix_heads128(const u128 *comp, const u64 *Uslot, u64 m, u64 *head)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x)
		head[u] = (u == 0 || comp[u] != comp[u - 1]) ? Uslot[u] : 0;
}
__global__ void __launch_bounds__(256)
k_ix_place(const u128 *comp, const u64 *vals, const u64 *Uslot, const u64 *rank, u64 m, u64 *SA, u64 *ISA, u64 *tied)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x) {
		const u64 i = vals[u], slot = Uslot[u];
		SA[slot] = i;
		ISA[i] = rank[u];
		const bool head = rank[u] == slot, next_head = u + 1 == m || comp[u + 1] != comp[u];
		tied[u] = (head && next_head) ? 0 : 1;
	}
}
__global__ void __launch_bounds__(256)
k_ix_compact(const u64 *vals, const u64 *Uslot, const u64 *rank, const u64 *tied, const u64 *where, u64 m, u64 u_at, u64 *Ngrp, u64 *Nslot, u64 *Nidx)
{
	for (u64 u = (u64)blockIdx.x * blockDim.x + threadIdx.x; u < m; u += (u64)gridDim.x * blockDim.x) {
		if (!tied[u]) continue;
		const u64 o = u_at + where[u];
		Ngrp[o] = rank[u]; Nslot[o] = Uslot[u]; Nidx[o] = vals[u];
	}
}
// largest e' <= e at which a group starts (slices are cut between groups); one thread
__global__ void k_ix_cut(const u64 *Ugrp, u64 lo, u64 e, u64 *out)
{
	while (e > lo + 1 && Ugrp[e] == Ugrp[e - 1]) --e;
	*out = e;
}

// ---------------------------------------------------------------------------------------------------------------------
// emit: BWT blocks and suffix-array samples
// ---------------------------------------------------------------------------------------------------------------------
// Row r of the sorted rotations (r = 0: the sentinel's own row) holds suffix SA'[r] with SA'[0] = n, SA'[r] = SA[r - 1];
// the BWT string leaves out the row of suffix 0 (`primary`).  One thread per 128-symbol block: symbols gathered from T,
// packed 16 per word (first symbol in the top bits) straight into the block's place in the file layout, counts kept.
__global__ void __launch_bounds__(256)
k_ix_bwt_blocks(const u64 *T, const u64 *SA, u64 n, u64 primary, uint32_t *bwt, unsigned int *cnt4)
{
	const u64 n_blocks = (n + 127) >> 7;
	for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (u64)gridDim.x * blockDim.x) {
		unsigned int c[4] = {0, 0, 0, 0};
		for (int wi = 0; wi < 8; ++wi) {
			const u64 w0 = (b << 7) + ((u64)wi << 4);
			if (w0 >= n) break;
			uint32_t word = 0;
			for (int k = 0; k < 16; ++k) {
				const u64 w = w0 + k;
				if (w >= n) break;
				const u64 r = w + (w >= primary);
				const u64 p = r ? SA[r - 1] : n;
				const int s = ix_base(T, p - 1);
				word |= (uint32_t)s << ((15 - k) << 1);
				c[0] += s == 0; c[1] += s == 1; c[2] += s == 2; c[3] += s == 3;
			}
			bwt[(b << 4) + 8 + wi] = word;
		}
		cnt4[b * 4 + 0] = c[0]; cnt4[b * 4 + 1] = c[1]; cnt4[b * 4 + 2] = c[2]; cnt4[b * 4 + 3] = c[3];
	}
}
// in: per-block counts of one symbol, widened for the scan
__global__ void __launch_bounds__(256)
k_ix_widen(const unsigned int *cnt4, int c, u64 n_blocks, u64 *out)
{
	for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (u64)gridDim.x * blockDim.x) out[b] = cnt4[b * 4 + c];
}
// running count of symbol c before block b -> the block's header; the block after the last one holds the totals
__global__ void __launch_bounds__(256)
k_ix_headers(const u64 *before, const unsigned int *cnt4, int c, u64 n_blocks, u64 tail_word, uint32_t *bwt)
{
	for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (u64)gridDim.x * blockDim.x) {
		u64 *hdr = (u64*)(bwt + (b << 4));
		hdr[c] = before[b];
		if (b + 1 == n_blocks) ((u64*)(bwt + tail_word))[c] = before[b] + cnt4[b * 4 + c];
	}
}
// out[j] = SA'[j * intv] for j < m; entry 0 is -1, as the loader leaves it (lib/aln/bwt.c:448-452)
__global__ void __launch_bounds__(256)
k_ix_sample(const u64 *SA, u64 intv, u64 m, u64 *out)
{
	for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (u64)gridDim.x * blockDim.x)
		out[j] = j ? SA[j * intv - 1] : ~0ull;
}

// ---------------------------------------------------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------------------------------------------------
struct MaxU64 { __host__ __device__ u64 operator()(u64 a, u64 b) const { return a > b ? a : b; } };

struct IxTemp {   // rocPRIM scratch, grown on demand
	DevBuf buf;
	int need(size_t n) { return buf.reserve(n); }
};
static int scan_max(IxTemp &tmp, hipStream_t st, u64 *in, u64 *out, size_t n)
{
	size_t tb = 0;
	IXCHK(rocprim::inclusive_scan(nullptr, tb, in, out, n, MaxU64(), st));
	RCCHK(tmp.need(tb));
	IXCHK(rocprim::inclusive_scan(tmp.buf.p, tb, in, out, n, MaxU64(), st));
	return BSX_OK;
}
static int scan_sum_excl(IxTemp &tmp, hipStream_t st, u64 *in, u64 *out, size_t n)
{
	size_t tb = 0;
	IXCHK(rocprim::exclusive_scan(nullptr, tb, in, out, 0ull, n, rocprim::plus<u64>(), st));
	RCCHK(tmp.need(tb));
	IXCHK(rocprim::exclusive_scan(tmp.buf.p, tb, in, out, 0ull, n, rocprim::plus<u64>(), st));
	return BSX_OK;
}
static int sort64(IxTemp &tmp, hipStream_t st, u64 *ki, u64 *ko, u64 *vi, u64 *vo, size_t n)
{
	size_t tb = 0;
	IXCHK(rocprim::radix_sort_pairs(nullptr, tb, ki, ko, vi, vo, n, 0u, 64u, st));
	RCCHK(tmp.need(tb));
	IXCHK(rocprim::radix_sort_pairs(tmp.buf.p, tb, ki, ko, vi, vo, n, 0u, 64u, st));
	return BSX_OK;
}
static int sort128(IxTemp &tmp, hipStream_t st, u128 *ki, u128 *ko, u64 *vi, u64 *vo, size_t n, unsigned int end_bit)
{
	size_t tb = 0;
	IXCHK(rocprim::radix_sort_pairs(nullptr, tb, ki, ko, vi, vo, n, 0u, end_bit, st));
	RCCHK(tmp.need(tb));
	IXCHK(rocprim::radix_sort_pairs(tmp.buf.p, tb, ki, ko, vi, vo, n, 0u, end_bit, st));
	return BSX_OK;
}
static inline int ix_grid(u64 n, int n_cu) { u64 g = (n + 255) / 256; u64 cap = (u64)n_cu * 32; return (int)std::max<u64>(1, std::min(g, cap)); }
static double ix_now() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static unsigned int bits_for(u64 v) { unsigned int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

int bsx_ix_build_fmi(hipStream_t st, int n_cu, const uint8_t *d_pac, int64_t l_pac, int parent, int dense_intv, int file_intv,
                     DevBuf *bwt_out, DevBuf *sa_out, bsx_fmi_t *meta, uint32_t *h_bwt, uint64_t *h_sa)
{
	const u64 n = (u64)l_pac << 1;
	if (l_pac < 64) return BSX_E_ARG;
	if (bits_for(n) + 1 >= IX_KEY2_BITS) return BSX_E_ARG;
	const bool trace = getenv("BSX_INDEX_TRACE") != nullptr || bsx_verbose >= 4;
	const double t_begin = ix_now();
	const u64 n_words = ((n + 31) >> 5) + 2;
	// batch / slice size: bounded so that the temporaries stay a small part of HBM
	u64 batch_cap = bsx_tune_is_set("index_batch") ? strtoull(bsx_tune_str("index_batch"), 0, 10) : ((u64)256 << 20);
	if (batch_cap < 1024) batch_cap = 1024;

	DevBuf T, SA, ISA, small, hist;
	IxTemp tmp;
	int rc = BSX_OK;
	struct Cleanup { std::vector<DevBuf*> v; ~Cleanup() { for (DevBuf *b : v) b->release(); } } cl;
	cl.v = {&T, &SA, &ISA, &small, &hist, &tmp.buf};
	RCCHK(T.reserve_exact(n_words * 8));
	RCCHK(small.reserve_exact(256));
	RCCHK(hist.reserve_exact((size_t)IX_BUCKETS * 8));
	IXCHK(hipMemsetAsync(small.p, 0, 256, st));
	IXCHK(hipMemsetAsync(hist.p, 0, (size_t)IX_BUCKETS * 8, st));
	u64 *d_small = (u64*)small.p;   // [0..3] symbol counts, [4] cursor, [5] cut
	hipLaunchKernelGGL(k_ix_text, dim3(ix_grid(n_words, n_cu)), dim3(256), 0, st, d_pac, (long long)l_pac, parent, (u64*)T.p, n_words, d_small);
	hipLaunchKernelGGL(k_ix_hist, dim3(n_cu * 4), dim3(256), 0, st, (const u64*)T.p, n, (u64*)hist.p);
	u64 sym[4];
	std::vector<u64> h_hist(IX_BUCKETS);
	IXCHK(hipMemcpyAsync(sym, d_small, 32, hipMemcpyDeviceToHost, st));
	IXCHK(hipMemcpyAsync(h_hist.data(), hist.p, (size_t)IX_BUCKETS * 8, hipMemcpyDeviceToHost, st));
	IXCHK(hipStreamSynchronize(st));
	IXCHK(hipGetLastError());
	if (sym[0] + sym[1] + sym[2] + sym[3] != n) return BSX_E_INTERNAL;
	meta->L2[0] = 0;
	for (int c = 0; c < 4; ++c) meta->L2[c + 1] = meta->L2[c] + sym[c];
	meta->seq_len = n;

	RCCHK(SA.reserve_exact(n * 8));
	RCCHK(ISA.reserve_exact(n * 8));
	u64 *d_SA = (u64*)SA.p, *d_ISA = (u64*)ISA.p;
	const u64 *d_T = (const u64*)T.p;

	// ---- round 0 ----
	u64 biggest = 0;
	for (int b = 0; b < IX_BUCKETS; ++b) biggest = std::max(biggest, h_hist[b]);
	const u64 bcap = std::max(std::min(batch_cap, n), biggest);
	DevBuf k0, k1, v0, v1, first, tied, where;
	cl.v.insert(cl.v.end(), {&k0, &k1, &v0, &v1, &first, &tied, &where});
	RCCHK(k0.reserve_exact(bcap * 16)); RCCHK(k1.reserve_exact(bcap * 16));   // sized for the 128-bit keys of the later rounds too
	RCCHK(v0.reserve_exact(bcap * 8)); RCCHK(v1.reserve_exact(bcap * 8));
	RCCHK(first.reserve_exact(bcap * 8)); RCCHK(tied.reserve_exact(bcap * 8)); RCCHK(where.reserve_exact(bcap * 8));
	// U grows batch by batch; ping-pong pair (grp, slot, idx) each
	struct UArr { DevBuf grp, slot, idx; u64 cap = 0;
		int grow(u64 want, hipStream_t s) {
			if (want <= cap) return BSX_OK;
			u64 nc = std::max<u64>(want + (want >> 1), (u64)1 << 20);
			DevBuf *bs[3] = {&grp, &slot, &idx};
			for (DevBuf *b : bs) {
				DevBuf nb;
				int rc = nb.reserve_exact(nc * 8);
				if (rc != BSX_OK) return rc;
				if (b->p && cap) { if (hipMemcpyAsync(nb.p, b->p, cap * 8, hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { nb.release(); return BSX_E_NODEVICE; } }
				b->release(); *b = nb;
			}
			cap = nc;
			return BSX_OK;
		} } U[2];
	cl.v.insert(cl.v.end(), {&U[0].grp, &U[0].slot, &U[0].idx, &U[1].grp, &U[1].slot, &U[1].idx});
	u64 um = 0;   // |U|
	{
		u64 base = 0;
		int b = 0, n_batches = 0;
		while (b < IX_BUCKETS) {
			u64 m = 0; int e = b;
			while (e < IX_BUCKETS && (m == 0 || m + h_hist[e] <= bcap)) { m += h_hist[e]; ++e; }
			if (m == 0) { b = e; continue; }
			IXCHK(hipMemsetAsync(d_small + 4, 0, 8, st));
			hipLaunchKernelGGL(k_ix_collect, dim3(n_cu * 8), dim3(256), 0, st, d_T, n, (unsigned)b, (unsigned)e, (u64*)k0.p, (u64*)v0.p, d_small + 4);
			RCCHK(sort64(tmp, st, (u64*)k0.p, (u64*)k1.p, (u64*)v0.p, (u64*)v1.p, (size_t)m));
			const int g = ix_grid(m, n_cu);
			hipLaunchKernelGGL(k_ix_heads64, dim3(g), dim3(256), 0, st, (const u64*)k1.p, m, (u64*)where.p);
			RCCHK(scan_max(tmp, st, (u64*)where.p, (u64*)first.p, (size_t)m));
			hipLaunchKernelGGL(k_ix_place0, dim3(g), dim3(256), 0, st, (const u64*)k1.p, (const u64*)v1.p, (const u64*)first.p, m, base, d_SA, d_ISA, (u64*)tied.p);
			RCCHK(scan_sum_excl(tmp, st, (u64*)tied.p, (u64*)where.p, (size_t)m));
			u64 last_w = 0, last_t = 0, got = 0;
			IXCHK(hipMemcpyAsync(&last_w, (u64*)where.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
			IXCHK(hipMemcpyAsync(&last_t, (u64*)tied.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
			IXCHK(hipMemcpyAsync(&got, d_small + 4, 8, hipMemcpyDeviceToHost, st));
			IXCHK(hipStreamSynchronize(st));
			if (got != m) { fprintf(stderr, "[bsx-index] batch of buckets [%d,%d): collected %llu suffixes, the histogram says %llu\n", b, e, got, m); return BSX_E_INTERNAL; }
			const u64 add = last_w + last_t;
			if (add) {
				RCCHK(U[0].grow(um + add, st));
				hipLaunchKernelGGL(k_ix_compact0, dim3(g), dim3(256), 0, st, (const u64*)v1.p, (const u64*)first.p, (const u64*)tied.p, (const u64*)where.p, m, base, um,
				                   (u64*)U[0].grp.p, (u64*)U[0].slot.p, (u64*)U[0].idx.p);
				um += add;
			}
			base += m; b = e; ++n_batches;
		}
		if (base != n) return BSX_E_INTERNAL;
		IXCHK(hipStreamSynchronize(st));
		IXCHK(hipGetLastError());
		if (trace) fprintf(stderr, "[bsx-index] %s: %llu suffixes, 32-mer sort in %d batches: %.2f s, %llu (%.2f %%) still tied\n", parent ? "parent" : "daughter", n, n_batches, ix_now() - t_begin, um, 100.0 * um / n);
	}

	// ---- doubling rounds ----
	int cur = 0;
	const unsigned int end_bit = IX_KEY2_BITS + bits_for(n);
	for (u64 h = 32; um > 0; h <<= 1) {
		if (h > n) return BSX_E_INTERNAL;
		const double t_r = ix_now();
		UArr &A = U[cur], &B = U[cur ^ 1];
		if (B.cap < um) { B.grp.release(); B.slot.release(); B.idx.release(); B.cap = 0; }   // nothing of its old content is needed
		RCCHK(B.grow(um, st));
		u64 new_um = 0, lo = 0;
		int n_slices = 0;
		while (lo < um) {
			u64 hi = std::min(um, lo + bcap);
			if (hi < um) {
				hipLaunchKernelGGL(k_ix_cut, dim3(1), dim3(1), 0, st, (const u64*)A.grp.p, lo, hi, d_small + 5);
				IXCHK(hipMemcpyAsync(&hi, d_small + 5, 8, hipMemcpyDeviceToHost, st));
				IXCHK(hipStreamSynchronize(st));
				if (hi <= lo + 1 && hi < um) { fprintf(stderr, "[bsx-index] a group of tied suffixes is larger than a slice (%llu): raise $BSX_INDEX_BATCH\n", bcap); return BSX_E_NOMEM; }
			}
			const u64 m = hi - lo;
			const int g = ix_grid(m, n_cu);
			const u64 *sl = (const u64*)A.slot.p + lo;
			hipLaunchKernelGGL(k_ix_key2, dim3(g), dim3(256), 0, st, (const u64*)A.grp.p + lo, (const u64*)A.idx.p + lo, m, (const u64*)d_ISA, n, h, (u128*)k0.p, (u64*)v0.p);
			RCCHK(sort128(tmp, st, (u128*)k0.p, (u128*)k1.p, (u64*)v0.p, (u64*)v1.p, (size_t)m, end_bit));
			hipLaunchKernelGGL(k_ix_heads128, dim3(g), dim3(256), 0, st, (const u128*)k1.p, sl, m, (u64*)where.p);
			RCCHK(scan_max(tmp, st, (u64*)where.p, (u64*)first.p, (size_t)m));   // first[u] = new rank of element u
			hipLaunchKernelGGL(k_ix_place, dim3(g), dim3(256), 0, st, (const u128*)k1.p, (const u64*)v1.p, sl, (const u64*)first.p, m, d_SA, d_ISA, (u64*)tied.p);
			RCCHK(scan_sum_excl(tmp, st, (u64*)tied.p, (u64*)where.p, (size_t)m));
			u64 last_w = 0, last_t = 0;
			IXCHK(hipMemcpyAsync(&last_w, (u64*)where.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
			IXCHK(hipMemcpyAsync(&last_t, (u64*)tied.p + (m - 1), 8, hipMemcpyDeviceToHost, st));
			IXCHK(hipStreamSynchronize(st));
			const u64 add = last_w + last_t;
			if (add) {
				hipLaunchKernelGGL(k_ix_compact, dim3(g), dim3(256), 0, st, (const u64*)v1.p, sl, (const u64*)first.p, (const u64*)tied.p, (const u64*)where.p, m, new_um,
				                   (u64*)B.grp.p, (u64*)B.slot.p, (u64*)B.idx.p);
				new_um += add;
			}
			lo = hi; ++n_slices;
		}
		IXCHK(hipStreamSynchronize(st));
		IXCHK(hipGetLastError());
		if (trace) fprintf(stderr, "[bsx-index]   depth %llu: %llu tied suffixes in %d slice(s) -> %llu, %.2f s\n", h, um, n_slices, new_um, ix_now() - t_r);
		um = new_um; cur ^= 1;
	}
	for (int k = 0; k < 2; ++k) { U[k].grp.release(); U[k].slot.release(); U[k].idx.release(); }
	k0.release(); k1.release(); v0.release(); v1.release(); tied.release();

	// ---- emit ----
	u64 isa0 = 0;
	IXCHK(hipMemcpyAsync(&isa0, d_ISA, 8, hipMemcpyDeviceToHost, st));
	IXCHK(hipStreamSynchronize(st));
	ISA.release();
	const u64 primary = isa0 + 1;
	meta->primary = primary;
	const u64 n_blocks = (n + 127) >> 7, n_occ = n_blocks + 1;
	const u64 bwt_words = ((n + 15) >> 4) + n_occ * 8;
	const u64 tail_word = bwt_words - 8;
	meta->bwt_size = bwt_words;
	RCCHK(bwt_out->reserve_exact((size_t)bwt_words * 4 + 64));
	IXCHK(hipMemsetAsync((char*)bwt_out->p + (size_t)bwt_words * 4, 0, 64, st));
	DevBuf cnt4;
	cl.v.push_back(&cnt4);
	RCCHK(cnt4.reserve_exact((size_t)n_blocks * 16));
	RCCHK(first.reserve_exact((size_t)n_blocks * 8)); RCCHK(where.reserve_exact((size_t)n_blocks * 8));
	const int gb = ix_grid(n_blocks, n_cu);
	hipLaunchKernelGGL(k_ix_bwt_blocks, dim3(gb), dim3(256), 0, st, d_T, (const u64*)d_SA, n, primary, (uint32_t*)bwt_out->p, (unsigned int*)cnt4.p);
	for (int c = 0; c < 4; ++c) {
		hipLaunchKernelGGL(k_ix_widen, dim3(gb), dim3(256), 0, st, (const unsigned int*)cnt4.p, c, n_blocks, (u64*)first.p);
		RCCHK(scan_sum_excl(tmp, st, (u64*)first.p, (u64*)where.p, (size_t)n_blocks));
		hipLaunchKernelGGL(k_ix_headers, dim3(gb), dim3(256), 0, st, (const u64*)where.p, (const unsigned int*)cnt4.p, c, n_blocks, tail_word, (uint32_t*)bwt_out->p);
	}
	// suffix-array samples: the device keeps a dense one, the files (and the host) the reference's 1-in-32
	const u64 n_dense = n / (u64)dense_intv + 1;
	RCCHK(sa_out->reserve_exact((size_t)n_dense * 8));
	hipLaunchKernelGGL(k_ix_sample, dim3(ix_grid(n_dense, n_cu)), dim3(256), 0, st, (const u64*)d_SA, (u64)dense_intv, n_dense, (u64*)sa_out->p);
	meta->sa_intv = file_intv;
	meta->n_sa = (n + (u64)file_intv) / (u64)file_intv;
	if (h_sa) {
		RCCHK(where.reserve_exact((size_t)meta->n_sa * 8));
		hipLaunchKernelGGL(k_ix_sample, dim3(ix_grid(meta->n_sa, n_cu)), dim3(256), 0, st, (const u64*)d_SA, (u64)file_intv, (u64)meta->n_sa, (u64*)where.p);
		IXCHK(hipMemcpyAsync(h_sa, where.p, (size_t)meta->n_sa * 8, hipMemcpyDeviceToHost, st));
	}
	if (h_bwt) IXCHK(hipMemcpyAsync(h_bwt, bwt_out->p, (size_t)bwt_words * 4, hipMemcpyDeviceToHost, st));
	IXCHK(hipStreamSynchronize(st));
	IXCHK(hipGetLastError());
	if (trace) fprintf(stderr, "[bsx-index] %s: done in %.2f s (primary %llu)\n", parent ? "parent" : "daughter", ix_now() - t_begin, primary);
	return rc;
}
