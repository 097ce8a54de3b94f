// k_seedt.hip -- K1+K2 over the table of k-mer intervals (seed_tab.hpp): the seeding kernel and the kernels that build the table, gfx950.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "seed_tab.hpp"
#include "kernels.h"
#include "wave.hpp"

// The FM blocks of a trip are fetched by the wave as a whole, as in k_seed.hip: the lanes post the block addresses they need, lane l loads
// piece l & 3 of request 16 r + (l >> 2) in round r (one full 64-byte line per four lanes), the pieces go back to their owners through LDS.
struct SeedtXchg {
	unsigned long long addr[128];
	uint4 slot[64 * 4];
};
__device__ __forceinline__ int seedt_mbcnt(unsigned long long m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
__device__ __forceinline__ DevBlock seedt_take(const SeedtXchg &X, int req)
{
	const int at = (req & 63) << 2, sw = (req >> 2) & 3;
	DevBlock b; b.v0 = X.slot[at + (0 ^ sw)]; b.v1 = X.slot[at + (1 ^ sw)]; b.v2 = X.slot[at + (2 ^ sw)]; b.v3 = X.slot[at + (3 ^ sw)];
	return b;
}

// One trip's memory: bwt_extend (lib/aln/bwt.c:278-293) of L.ik for the lanes with kind == SQ_FM, the table entry L.tab_idx for the
// lanes with kind == SQ_TAB (one 16-byte load of their own, in flight together with the blocks).  Every lane of the wave takes part.
__device__ __forceinline__ DevIntv seedt_fetch_wave(int kind, const DevIndex &ix, SeedLane2 &L, SeedtXchg &X)
{
	const int lane = (int)(threadIdx.x & 63), piece = lane & 3, sub = lane >> 2;
	const bool need = kind == SQ_FM;
	uint4 tv = make_uint4(0, 0, 0, 0);
	if (kind == SQ_TAB) tv = (L.parent ? ix.tab.t[1] : ix.tab.t[0])[L.tab_idx];
	const int which = L.ext_which ? !L.parent : L.parent;
	const uint64_t primary = dev_ix_primary(ix, which);
	const uint32_t *bwt = dev_ix_bwt(ix, which);
	const int is_back = L.ext_back, c = L.ext_c;
	const uint64_t xa = is_back ? L.ik.x0 : L.ik.x1;
	const uint64_t xb = is_back ? L.ik.x1 : L.ik.x0;
	const uint64_t x2 = L.ik.x2;
	const uint64_t NEG1 = ~0ull;
	const uint64_t k = xa - 1, l = xa - 1 + x2;
	const uint64_t ka = k - (k >= primary), la = l - (l >= primary);
	const bool kv = k != NEG1, lv = l != NEG1;
	const bool same = kv && lv && (ka >> 7) == (la >> 7);
	const bool r0 = need, r1 = need && !same;
	const unsigned long long m0 = __ballot(r0), m1 = __ballot(r1);
	const int n0 = __popcll(m0), n = n0 + __popcll(m1);
	const int i0 = seedt_mbcnt(m0), i1 = n0 + seedt_mbcnt(m1);
	if (r0) X.addr[i0] = (unsigned long long)(bwt + (((kv ? ka : 0) >> 7) << 4));
	if (r1) X.addr[i1] = (unsigned long long)(bwt + (((lv ? la : 0) >> 7) << 4));
	WAVE_SYNC();
	const int rounds = (n + 15) >> 4;
	uint64_t tk[4] = {0, 0, 0, 0}, tl[4] = {0, 0, 0, 0};
#define SEEDT_COUNT(B_, pos_, valid_, t_) do { uint32_t a_, c_, g_, t4_; dev_planes_count4(B_, (int)((pos_) & 127), a_, c_, g_, t4_); \
		t_[0] = (valid_) ? ((uint64_t)B_.v0.y << 32 | B_.v0.x) + a_ : 0; t_[1] = (valid_) ? ((uint64_t)B_.v0.w << 32 | B_.v0.z) + c_ : 0; \
		t_[2] = (valid_) ? ((uint64_t)B_.v1.y << 32 | B_.v1.x) + g_ : 0; t_[3] = (valid_) ? ((uint64_t)B_.v1.w << 32 | B_.v1.z) + t4_ : 0; } while (0)
	uint4 V[8];
#pragma unroll
	for (int r = 0; r < 8; ++r) {
		V[r] = make_uint4(0, 0, 0, 0);
		if (r < rounds) {
			const int req = r * 16 + sub;
			if (req < n) V[r] = reinterpret_cast<const uint4*>(X.addr[req])[piece];
		}
	}
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		if (h * 64 < n) {
#pragma unroll
			for (int rr = 0; rr < 4; ++rr) {
				const int r = 4 * h + rr, req = r * 16 + sub;
				if (req < n) X.slot[((req & 63) << 2) + (piece ^ ((req >> 2) & 3))] = V[r];
			}
			WAVE_SYNC();
			if (h == 0 && r0) {
				const DevBlock B = seedt_take(X, i0);
				SEEDT_COUNT(B, ka, kv, tk);
				if (same) SEEDT_COUNT(B, la, lv, tl);
			}
			if (r1 && (i1 >> 6) == h) { const DevBlock B = seedt_take(X, i1); SEEDT_COUNT(B, la, lv, tl); }
			WAVE_SYNC();
		}
	}
	DevIntv o;
	if (kind == SQ_TAB) {
		SeedEnt e; e.x0 = tv.x; e.x1 = tv.y; e.x2 = tv.z; e.hi = tv.w;
		o = seed_unpack(e); o.info = 0;
		++L.n_look;
		return o;
	}
	if (need) { if (same) ++L.n_fast; else ++L.n_slow; }
	const uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1], s0 = tl[0] - tk[0];
	const uint64_t b3 = xb + ((xa <= primary && xa + x2 - 1 >= primary) ? 1 : 0);
	const uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const uint64_t tkc = c == 3 ? tk[3] : c == 2 ? tk[2] : c == 1 ? tk[1] : tk[0];
	const uint64_t na = dev_ix_L2(ix, which, c) + 1 + tkc;
	const uint64_t nb = c == 3 ? b3 : c == 2 ? b2 : c == 1 ? b1 : b0;
	const uint64_t ns = c == 3 ? s3 : c == 2 ? s2 : c == 1 ? s1 : s0;
	o.x0 = is_back ? na : nb; o.x1 = is_back ? nb : na; o.x2 = ns; o.info = 0;
	return o;
}

// The reads of a chunk as the table's digits (seed2_digit_of: converted for the index each strand search runs on), two bits a base, sixteen bases
// a word, SEEDT_WPT words per strand search: what a lane copies into LDS when it takes the strand search.  A thread per word.
#define SEEDT_WPT 12    // 192 bases; longer reads are read where they lie
__global__ void __launch_bounds__(256)
k_seedt_pack(const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, uint32_t *qpack)
{
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	const int t = (int)(i / SEEDT_WPT), w = (int)(i % SEEDT_WPT);
	if (t >= n_tasks) return;
	const int len = tasks[t].len, parent = tasks[t].parent;
	if (len > SEEDT_WPT * 16 || w * 16 >= len) return;
	const uint8_t *q = reads + tasks[t].qoff + w * 16;
	uint32_t pk = 0;
	for (int b = 0; b < 16 && w * 16 + b < len; ++b) pk |= seed2_digit_of(q[b], parent) << (b << 1);
	qpack[(size_t)t * SEEDT_WPT + w] = pk;
}

// One lane = one strand search at a time; a lane that finishes one takes the next off the global cursor (`quota` per lane, 0 = until
// the cursor runs out), so a wave stays full until the queue drains.  A trip of the wave loop: the lanes whose strand search is done
// publish it and take another (not on every trip: the wave waits until a few lanes are at that point), every lane runs its machine up
// to its next request (seed_tab.hpp), the wave fetches all requests together, the lanes consume theirs.  Launch geometry, slabs and
// counters are those of k_seed (k_seed.hip).
template <int OCC>
__global__ void __launch_bounds__(64, OCC)
k_seedt(DevIndex ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, SeedParams P,
        DevIntv *scratch, int list_cap, int mem_cap,
        DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
        long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
        int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof, unsigned int cold_mask, int cold_lanes, const uint32_t *qpack,
        unsigned long long direct_off)
{
	int slab = 0;
	if ((threadIdx.x & 63) == 0) {
		unsigned int h = (unsigned int)(blockIdx.x % (unsigned int)n_slabs);
		while (atomicCAS(&slab_busy[h], 0u, 1u) != 0u) h = h + 1 == (unsigned int)n_slabs ? 0 : h + 1;
		slab = (int)h;
	}
	slab = __builtin_amdgcn_readfirstlane(__shfl(slab, 0));
	const size_t slab_bytes = (size_t)64 * ((size_t)mem_cap * sizeof(DevIntv) + (size_t)list_cap * sizeof(SeedEnt));
	char *slab_base = reinterpret_cast<char*>(scratch) + (size_t)slab * slab_bytes;
	const int K = ix.tab.K;
	SeedLane2 L;
	L.stride = 64; L.lane = (int)(threadIdx.x & 63);
	// the SMEMs of a strand search: in the wave's slab until it is done, then copied behind the others in `out` (direct_off == ~0) -- or, when the
	// caller has room for mem_cap entries per strand search, written where they stay: out[direct_off + task * mem_cap ...] (no copy, no cursor)
	const bool direct = direct_off != ~0ull;
	DevIntv *const slab_mem = reinterpret_cast<DevIntv*>(slab_base) + (threadIdx.x & 63);
	L.mem = slab_mem; L.mstride = direct ? (uint32_t)sizeof(DevIntv) : 64u * (uint32_t)sizeof(DevIntv);
	L.bufA = reinterpret_cast<SeedEnt*>(slab_base + (size_t)64 * mem_cap * sizeof(DevIntv));
	L.list_cap = list_cap; L.mem_cap = mem_cap;
	__shared__ uint32_t s_read[SEEDT_WPT][64];
	__shared__ SeedtXchg X;
	uint32_t *my_read = &s_read[0][threadIdx.x & 63];
	L.qlds = nullptr; L.q = reads; L.len = 0; L.parent = 0;
	L.state = ST_DONE;
	L.n_slow = L.n_fast = L.n_look = 0;
	L.ik.x0 = L.ik.x1 = L.ik.x2 = L.ik.info = 0; L.ext_back = L.ext_c = L.ext_which = 0; L.tab_idx = 0;
	L.mem_n = 0; L.overflow = 0; L.pw = 0; L.key = 0; L.ka = L.kb = 0;
	int task = -1, retired = 0, taken = 0, trips = 0, budget = 0;
	uint32_t tot_slow = 0, tot_fast = 0, tot_look = 0, tot_over = 0;   // (tot_over: strand searches this lane left with a negative count)
	unsigned int trip = 0;
	const int hist = prof & 2; prof &= 1;
	long long pc_t0 = prof ? clock64() : 0, pc_cold = 0, pc_hot = 0, pc_fetch = 0, pc_post = 0; unsigned int pc_cold_n = 0;
	unsigned long long pc_req = 0;
	for (;;) {
		const bool idle = !retired && L.state == ST_DONE;
		const unsigned long long im = __ballot(idle);
		++trip;
		const bool go = idle && ((trip & cold_mask) == 0 || __popcll(im) > cold_lanes || __popcll(im) + __popcll(__ballot(retired != 0)) == 64);
		long long pc_c0 = 0;
		if (prof && __ballot(go)) { pc_c0 = clock64(); ++pc_cold_n; }
		if (go) {
			if (task >= 0) { // publish the finished strand search
				unsigned long long base = direct ? direct_off + (unsigned long long)task * (unsigned long long)mem_cap : 0;
				int n = L.mem_n;
				if (!direct && n > 0 && !L.overflow) {
					base = atomicAdd(out_cursor, (unsigned long long)n);
					if (base + n <= out_cap) for (int k = 0; k < n; ++k) out[base + k] = seed2_mem_at(L, k);
					else L.overflow = 1;
				}
				task_off[task] = (long long)base;
				task_n[task] = L.overflow ? -n - 1 : n;   // any negative count: seed this strand search again
				tot_over += L.overflow ? 1u : 0u;
				if (hist) atomicAdd(&counters[60 + (trips > 0 ? 32 - __clz(trips) : 0)], 1ull);   // ($BSX_PHASES: requests per strand search of the second pass, by power of two)
				tot_slow += L.n_slow; tot_fast += L.n_fast; tot_look += L.n_look;
				task = -1;
			}
			for (;;) {
				if (quota && taken >= quota) { retired = 1; break; }
				const unsigned int t = atomicAdd(task_cursor, 1u);
				if (t >= (unsigned int)n_tasks) { retired = 1; break; }
				++taken;
				task = (int)t;
				L.q = reads + tasks[t].qoff; L.len = tasks[t].len; L.parent = tasks[t].parent;
				L.qlds = nullptr;
				if (direct) L.mem = out + direct_off + (unsigned long long)t * (unsigned long long)mem_cap;
				seed2_lane_begin(L);
				trips = 0;
				budget = trip_budget * ((L.len + 255) >> 8);
				if (L.len < P.min_seed_len || L.len + 1 > list_cap) { // too short to seed (memchain.c:279) / cannot fit: an empty result
					task_off[t] = 0; task_n[t] = L.len + 1 > list_cap ? -1 : 0;
					tot_over += L.len + 1 > list_cap ? 1u : 0u;
					task = -1;
					continue;
				}
				if (qpack && L.len <= SEEDT_WPT * 16) {
					const uint32_t *src = qpack + (size_t)t * SEEDT_WPT;
					for (int w = 0; w * 16 < L.len; ++w) my_read[w << 6] = src[w];
					L.qlds = my_read;
				}
				break;
			}
			if (retired) L.state = ST_DONE;
		}
		if (prof && pc_c0) pc_cold += clock64() - pc_c0;
		if (__all(retired)) break;
		const long long pc_h0 = prof ? clock64() : 0;
		int kind = SQ_NONE;
		if (!retired && L.state != ST_DONE) kind = seed2_advance(L, ix, P, K);
		if (prof) pc_hot += clock64() - pc_h0;
		if (__ballot(kind > 0) == 0) continue;
		const long long pc_f0 = prof ? clock64() : 0;
		if (prof) pc_req += (unsigned long long)__popcll(__ballot(kind > 0));
		const DevIntv ok = seedt_fetch_wave(kind, ix, L, X);
		const long long pc_p0 = prof ? clock64() : 0;
		if (prof) pc_fetch += pc_p0 - pc_f0;
		if (kind > 0) {
			seed2_post(L, ok, ix, P, K);
			// a strand search whose lists no longer fit, or that has made `budget` requests (low-complexity reads), is abandoned at once
			// and seeded again on the side stream with longer lists and no budget (k_seed.hip has the reasoning)
			if (budget && ++trips > budget) L.overflow = 1;
			if (L.overflow) L.state = ST_DONE;
		}
		if (prof) pc_post += clock64() - pc_p0;
	}
	for (int off = 32; off > 0; off >>= 1) { tot_slow += __shfl_down(tot_slow, off); tot_fast += __shfl_down(tot_fast, off); tot_look += __shfl_down(tot_look, off); tot_over += __shfl_down(tot_over, off); }
	if ((threadIdx.x & 63) == 0) {
		if (prof) {
			atomicAdd(&counters[48], (unsigned long long)(clock64() - pc_t0)); atomicAdd(&counters[49], (unsigned long long)pc_cold);
			atomicAdd(&counters[51], (unsigned long long)trip); atomicAdd(&counters[52], (unsigned long long)pc_cold_n);
			atomicAdd(&counters[121], (unsigned long long)pc_hot); atomicAdd(&counters[122], (unsigned long long)pc_fetch); atomicAdd(&counters[123], (unsigned long long)pc_post);
			atomicAdd(&counters[124], pc_req);
		}
		atomicAdd(&counters[0], 2ull * tot_slow); atomicAdd(&counters[1], (unsigned long long)tot_fast); atomicAdd(&counters[120], (unsigned long long)tot_look);
		if (tot_over) atomicAdd(&counters[119], (unsigned long long)tot_over);   // what the caller has to seed again: read back instead of every count
		__threadfence();
		atomicExch(&slab_busy[slab], 0u);
	}
}

void launch_seedt(hipStream_t st, int grid, const DevIndex &ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, const SeedParams &P,
                  DevIntv *scratch, int list_cap, int mem_cap, DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
                  long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
                  int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof, uint32_t *qpack, unsigned long long direct_off)
{
	// lanes whose strand search is done wait for company before the wave publishes and hands out new ones: every (cold_mask + 1)-th trip, or
	// when more than cold_lanes wait (every 8th trip / 16 lanes: swept in round 4)
	const unsigned int cold_mask = 8u - 1u;
	const int cold_lanes = 16;
	if (qpack) {
		const long long nw = (long long)n_tasks * SEEDT_WPT;
		hipLaunchKernelGGL(k_seedt_pack, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, reads, tasks, n_tasks, qpack);
	}
	hipLaunchKernelGGL((k_seedt<3>), dim3(grid * 4), dim3(64), 0, st, /* `grid` counts groups of four waves */ ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap,
	                   out, out_cap, out_cursor, task_off, task_n, task_cursor, counters, quota, slab_busy, n_slabs, trip_budget, prof, cold_mask, cold_lanes, (const uint32_t*)qpack, direct_off);
}
// the interval lists of the listed strand searches copied one behind the other (dst_off[j]: where list j starts): what the caller downloads
// when it has to take strand searches back (with the lists where k_seedt left them, a gigabyte apart, one bulk copy is not an option)
__global__ void __launch_bounds__(64)
k_gather_lists(const DevIntv *src, const long long *off, const int *cnt, const long long *which, const long long *dst_off, long long n_list, DevIntv *dst)
{
	const long long j = blockIdx.x;
	if (j >= n_list) return;
	const long long i = which[j];
	const int n = cnt[i];
	for (int k = (int)threadIdx.x; k < n; k += 64) dst[dst_off[j] + k] = src[off[i] + k];
}
void launch_gather_lists(hipStream_t st, const DevIntv *src, const long long *off, const int *cnt, const long long *which, const long long *dst_off, long long n_list, DevIntv *dst)
{
	if (n_list > 0) hipLaunchKernelGGL(k_gather_lists, dim3((unsigned)n_list), dim3(64), 0, st, src, off, cnt, which, dst_off, n_list, dst);
}
size_t seedt_pack_bytes(long long n_tasks) { return (size_t)n_tasks * SEEDT_WPT * 4 + 64; }

// ---- the table: level 1 = bwt_set_intv of the three letters, level L + 1 from level L, a thread per entry of level L (one bwt_2occ4 of
// the complementary index gives its three children)
__global__ void k_seedtab_first(DevIndex ix, int parent, SeedEnt *T)
{
	if (threadIdx.x < 3) { DevIntv ik; seed_set_intv(ix, parent, seed_letter(threadIdx.x, parent), ik); T[threadIdx.x] = seed_pack(ik); }
}
__global__ void __launch_bounds__(256)
k_seedtab_level(DevIndex ix, int parent, SeedEnt *T, unsigned long long at, unsigned long long nx, unsigned long long n_at)
{
	const unsigned long long key = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (key >= n_at) return;
	const DevIntv p = seed_unpack(T[at + key]);
	if (p.x2 == 0) return;   // (the table starts out as zeroes)
	const DevFmi o = dev_fmi_pick(ix, !parent);
	uint64_t tk[4], tl[4];
	dev_2occ4_planes(o, p.x1 - 1, p.x1 - 1 + p.x2, tk, tl);
	SeedEnt ch[3];
	seed_tab_children(p, tk, tl, o.primary, o.L2, parent, ch);
	T[nx + key * 3] = ch[0]; T[nx + key * 3 + 1] = ch[1]; T[nx + key * 3 + 2] = ch[2];
}
int seedtab_build(hipStream_t st, const DevIndex &ix, int parent, int K, void *table)
{
	SeedEnt *T = (SeedEnt*)table;
	if (hipMemsetAsync(T, 0, (size_t)seed_tab_entries(K) * sizeof(SeedEnt), st) != hipSuccess) return BSX_E_NODEVICE;
	hipLaunchKernelGGL(k_seedtab_first, dim3(1), dim3(64), 0, st, ix, parent, T);
	unsigned long long pw = 3;
	for (int l = 1; l < K; ++l, pw *= 3) {
		const unsigned long long at = (pw - 3) >> 1, nx = (pw * 3 - 3) >> 1;
		hipLaunchKernelGGL(k_seedtab_level, dim3((unsigned)((pw + 255) / 256)), dim3(256), 0, st, ix, parent, T, at, nx, pw);
	}
	return hipGetLastError() == hipSuccess ? BSX_OK : BSX_E_NODEVICE;
}
