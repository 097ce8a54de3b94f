// k_seed.hip -- K1+K2 (FM-index SMEM seeding) and K3 (suffix-array lookup) kernels, gfx950.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "seed_core.hpp"
#include "kernels.h"
#include "tune.h"

#include "wave.hpp"

// the read in LDS: 24 words = 192 bases per lane, 6 KB per wave
// One wave per workgroup: a workgroup gives its registers and LDS back when its LAST wave ends, and a wave ends when the longest of its 64
// strand searches does -- with four waves that was the longest of 256 while the slots of the three finished waves stayed taken
#ifndef SEED_WPB
#define SEED_WPB 1
#endif

// The FM blocks of one trip are fetched by the wave as a whole.  A lane that reads its own 64-byte block issues four 16-byte
// loads, each of which is one request for a line no other lane of the instruction shares: 64 lines per instruction, and the
// line has to survive in L1 until the fourth load (it often does not: tools/ubench/gather64.hip, 2.0 TB/s for two blocks per
// lane against 3.5 TB/s when four adjacent lanes read one block).  So the lanes post the block addresses they need, in
// request order; lane l then loads piece l&3 of request 16r + (l>>2) in round r -- one full 64-byte line per four lanes, every
// line touched by exactly one instruction -- and the pieces go back to their owners through LDS.
struct SeedXchg {
	unsigned long long addr[128];   // block addresses of this trip's requests (<= two per lane)
	uint4 slot[64 * 4];             // blocks of 64 requests on their way back: piece p of request r at slot[(r&63)*4 + (p ^ (r>>2 & 3))]
};

__device__ __forceinline__ int seed_mbcnt(unsigned long long m)
{
	return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
__device__ __forceinline__ DevBlock seed_take(const SeedXchg &X, int req)
{
	const int at = (req & 63) << 2, sw = (req >> 2) & 3;
	DevBlock b; b.v0 = X.slot[at + (0 ^ sw)]; b.v1 = X.slot[at + (1 ^ sw)]; b.v2 = X.slot[at + (2 ^ sw)]; b.v3 = X.slot[at + (3 ^ sw)];
	return b;
}

// bwt_extend (lib/aln/bwt.c:278-293) of L.ext_in for the lanes with `need`; every lane of the wave takes part in the loads.
__device__ __forceinline__ DevIntv seed_extend_wave(bool need, const DevIndex &ix, const SeedLane &L, SeedXchg &X, uint32_t &n_slow, uint32_t &n_fast)
{
	const int lane = (int)(threadIdx.x & 63), piece = lane & 3, sub = lane >> 2;
	const int which = L.ext_which ? !L.parent : L.parent;
	const uint64_t primary = dev_ix_primary(ix, which);
	const uint32_t *bwt = dev_ix_bwt(ix, which);
	const int is_back = L.ext_back, c = L.ext_c;
	const uint64_t xa = is_back ? L.ext_in.x0 : L.ext_in.x1;   // x[!is_back]
	const uint64_t xb = is_back ? L.ext_in.x1 : L.ext_in.x0;   // x[is_back]
	const uint64_t x2 = L.ext_in.x2;
	// bwt_2occ4 (lib/aln/bwt.c:204-236) at k = xa - 1 and l = xa - 1 + x2
	const uint64_t NEG1 = ~0ull;
	const uint64_t k = xa - 1, l = xa - 1 + x2;
	const uint64_t ka = k - (k >= primary), la = l - (l >= primary);
	const bool kv = k != NEG1, lv = l != NEG1;
	const bool same = kv && lv && (ka >> 7) == (la >> 7);
	const bool r0 = need, r1 = need && !same;
	const unsigned long long m0 = __ballot(r0), m1 = __ballot(r1);
	const int n0 = __popcll(m0), n = n0 + __popcll(m1);
	const int i0 = seed_mbcnt(m0), i1 = n0 + seed_mbcnt(m1);
	if (r0) X.addr[i0] = (unsigned long long)(bwt + (((kv ? ka : 0) >> 7) << 4));   // k == -1 reads block 0 and is discarded
	if (r1) X.addr[i1] = (unsigned long long)(bwt + (((lv ? la : 0) >> 7) << 4));
	WAVE_SYNC();
	const int rounds = (n + 15) >> 4;
	uint64_t tk[4] = {0, 0, 0, 0}, tl[4] = {0, 0, 0, 0};
	// block -> the four cumulative counts at position pos_ (bwt_occ4): counted as soon as a block is taken, so that only one is held in registers
#define SEED_COUNT(B_, pos_, valid_, t_) do { uint32_t a_, c_, g_, t4_; dev_planes_count4(B_, (int)((pos_) & 127), a_, c_, g_, t4_); \
		t_[0] = (valid_) ? ((uint64_t)B_.v0.y << 32 | B_.v0.x) + a_ : 0; t_[1] = (valid_) ? ((uint64_t)B_.v0.w << 32 | B_.v0.z) + c_ : 0; \
		t_[2] = (valid_) ? ((uint64_t)B_.v1.y << 32 | B_.v1.x) + g_ : 0; t_[3] = (valid_) ? ((uint64_t)B_.v1.w << 32 | B_.v1.z) + t4_ : 0; } while (0)
	// (FM blocks straight into LDS -- global_load_lds_dwordx4 -- were measured in round 3: 221 against 209 ms, and 363 at four waves per SIMD)
	uint4 V[8];
	{
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			V[r] = make_uint4(0, 0, 0, 0);
			if (r < rounds) {
				const int req = r * 16 + sub;
				if (req < n) V[r] = reinterpret_cast<const uint4*>(X.addr[req])[piece];
			}
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
			if (h == 0 && r0) { // n0 <= 64: every first block travels in the first half
				const DevBlock B = seed_take(X, i0);
				SEED_COUNT(B, ka, kv, tk);
				if (same) SEED_COUNT(B, la, lv, tl);
			}
			if (r1 && (i1 >> 6) == h) { const DevBlock B = seed_take(X, i1); SEED_COUNT(B, la, lv, tl); }
			WAVE_SYNC();
		}
	}
	if (same) ++n_fast; else ++n_slow;
	const uint64_t s3 = tl[3] - tk[3], s2 = tl[2] - tk[2], s1 = tl[1] - tk[1], s0 = tl[0] - tk[0];
	const uint64_t b3 = xb + ((xa <= primary && xa + x2 - 1 >= primary) ? 1 : 0);
	const uint64_t b2 = b3 + s3, b1 = b2 + s2, b0 = b1 + s1;
	const uint64_t tkc = c == 3 ? tk[3] : c == 2 ? tk[2] : c == 1 ? tk[1] : tk[0];
	const uint64_t na = dev_ix_L2(ix, which, c) + 1 + tkc;
	const uint64_t nb = c == 3 ? b3 : c == 2 ? b2 : c == 1 ? b1 : b0;
	const uint64_t ns = c == 3 ? s3 : c == 2 ? s2 : c == 1 ? s1 : s0;
	DevIntv o;
	o.x0 = is_back ? na : nb; o.x1 = is_back ? nb : na; o.x2 = ns; o.info = 0;
	return o;
}

// One lane = one strand search at a time; lanes pull the next task from a global cursor as soon
// as they finish (reads differ a lot in seeding work), so a wave stays full until the queue drains.
// Every trip of the outer loop issues the FM-block gathers of all 64 lanes together.
// A lane retires after `quota` tasks (0 = never): workgroups then have a bounded life and the launch is
// many more workgroups than fit on the chip, which lets kernels of a higher-priority stream (the back half of
// the previous chunk) get compute units while this one is running.  Scratch slabs are therefore not tied to
// the workgroup index: each wave takes a free slab and gives it back when it exits.
template <int OCC, int LDS_WORDS>
__global__ void __launch_bounds__(64 * SEED_WPB, OCC)
k_seed(DevIndex ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, SeedParams P,
       DevIntv *scratch, int list_cap, int mem_cap,
       DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
       long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
       int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof, unsigned int cold_mask, int cold_lanes)
{
	// per-wave slab, lane-interleaved: entry i of lane l sits at slab[i*64 + l], so the 64 lanes'
	// accesses to the same list position form one contiguous 2 KB run (coalesced, one TLB page)
	int slab = 0;
	if ((threadIdx.x & 63) == 0) {
		unsigned int h = (unsigned int)((blockIdx.x * (unsigned)SEED_WPB + (threadIdx.x >> 6)) % (unsigned int)n_slabs);
		while (atomicCAS(&slab_busy[h], 0u, 1u) != 0u) h = h + 1 == (unsigned int)n_slabs ? 0 : h + 1;
		slab = (int)h;
	}
	slab = __builtin_amdgcn_readfirstlane(__shfl(slab, 0));   // wave-uniform, and known to be: the slab's address stays in scalar registers
	const size_t wave_id = (size_t)slab;
	// a slab: per lane mem_cap SMEM records of 32 bytes, then one list of list_cap 16-byte entries, both lane-interleaved
	const size_t slab_bytes = (size_t)64 * ((size_t)mem_cap * sizeof(DevIntv) + (size_t)list_cap * sizeof(SeedEnt));
	char *slab_base = reinterpret_cast<char*>(scratch) + wave_id * slab_bytes;
	SeedLane L;
	L.stride = 64; L.lane = (int)(threadIdx.x & 63);
	L.mem = reinterpret_cast<DevIntv*>(slab_base);
	L.bufA = reinterpret_cast<SeedEnt*>(slab_base + (size_t)64 * mem_cap * sizeof(DevIntv));
	L.list_cap = list_cap; L.mem_cap = mem_cap;
	// the read, bisulfite-converted and packed 8 bases/word, lane-interleaved in LDS (reads of up to LDS_WORDS * 8 bases; longer ones are read where they lie)
	__shared__ uint32_t s_read[SEED_WPB][LDS_WORDS][64];
	__shared__ SeedXchg s_xchg[SEED_WPB];
	SeedXchg &X = s_xchg[threadIdx.x >> 6];
	uint32_t *my_read = &s_read[threadIdx.x >> 6][0][threadIdx.x & 63];
	L.qlds = nullptr;
	L.state = SD_DONE;
	L.n_slow = L.n_fast = 0;
	int task = -1, retired = 0, taken = 0, trips = 0, budget = 0;
	uint32_t tot_slow = 0, tot_fast = 0, tot_over = 0;

	unsigned int trip = 0;
	// $BSX_PHASES: where a wave's cycles go (u64 slots 48..: wave cycles, in the full machine, publishing, trips, full-machine passes)
	long long pc_t0 = prof ? clock64() : 0, pc_cold = 0, pc_pub = 0; unsigned int pc_cold_n = 0;
	for (;;) {
		// Every trip: the short machine (forward walk, backward sweep, LAST-like walk).  A lane that reaches any other state
		// waits for the full machine, which the wave runs every fourth trip, or at once when 25 lanes wait or no lane has an
		// extension to do: its code (pass control, SMEM prologue/epilogue, publishing, fetching and packing the next read) is
		// several times longer than a trip's, and a wave pays for every state any of its lanes is in.  (Measured: 180 -> 174 ms,
		// and the same with the full machine on every trip: what helps is that the common states leave through the short copy.)
		int need = 0;
		if (!retired) need = seed_advance_t<true>(L, ix, P) == 1;
		const bool cold = !retired && !need;
		const unsigned long long cm = __ballot(cold);
		++trip;
		const bool go_cold = cold && ((trip & cold_mask) == 0 || __popcll(cm) > cold_lanes || __ballot(need) == 0);
		long long pc_c0 = 0;
		if (prof && __ballot(go_cold)) { pc_c0 = clock64(); ++pc_cold_n; }
		if (go_cold) {
			for (;;) {
				if (L.state == SD_DONE) {
					if (task >= 0) { // publish the finished task
						const long long pc_p0 = prof ? clock64() : 0;
						unsigned long long base = 0;
						int n = L.mem_n;
						if (n > 0 && !L.overflow) {
							base = atomicAdd(out_cursor, (unsigned long long)n);
							if (base + n <= out_cap) for (int k = 0; k < n; ++k) out[base + k] = seed_mem_at(L, k);
							else L.overflow = 1;
						}
						task_off[task] = (long long)base;
						task_n[task] = L.overflow ? -n - 1 : n;   // any negative count: seed this strand search again
						tot_over += L.overflow ? 1u : 0u;
						tot_slow += L.n_slow; tot_fast += L.n_fast;
						task = -1;
						if (prof) pc_pub += clock64() - pc_p0;
					}
					if (quota && taken >= quota) { retired = 1; break; }
					unsigned int t = atomicAdd(task_cursor, 1u);
					if (t >= (unsigned int)n_tasks) { retired = 1; break; }
					++taken;
					task = (int)t;
					L.q = reads + tasks[t].qoff; L.len = tasks[t].len; L.parent = tasks[t].parent;
					L.qlds = nullptr;
					if (L.len <= LDS_WORDS * 8) {
						for (int w = 0; w * 8 < L.len; ++w) {
							uint32_t pk = 0;
							for (int b = 0; b < 8 && w * 8 + b < L.len; ++b) {
								int v = L.q[w * 8 + b];
								v = L.parent ? (v == 1 ? 3 : v) : (v == 2 ? 0 : v);
								pk |= (uint32_t)v << (b << 2);
							}
							my_read[w << 6] = pk;
						}
						L.qlds = my_read;
					}
					seed_lane_begin(L);
					trips = 0;
					budget = trip_budget * ((L.len + 255) >> 8);   // per 256 bases: a long read is not a runaway
					if (L.len < P.min_seed_len || L.len + 1 > list_cap) { // too short to seed (memchain.c:279) / cannot fit
						if (L.len + 1 > list_cap) L.overflow = 1;
						L.state = SD_DONE;
						continue;
					}
				}
				need = seed_advance(L, ix, P);
				if (need) break;
			}
		}
		if (prof && pc_c0) pc_cold += clock64() - pc_c0;
		if (__all(retired)) break;
		if (__ballot(need) == 0) continue;
		const DevIntv ok = seed_extend_wave(need, ix, L, X, L.n_slow, L.n_fast);
		if (need) {
			seed_post(L, ok, P);
			// a strand search whose lists no longer fit is abandoned at once: its result is discarded and it is seeded again with
			// longer lists, so finishing it here would only keep this wave (and the kernel's tail) alive for nothing
			// The same goes for a strand search that has taken `trip_budget` extensions (a few in a hundred thousand do: low-complexity
			// reads whose lists stay just inside their bounds).  The kernel lasts at least as long as its longest dependent chain, and
			// one lane's chain of 20 k extensions is a third of the time the whole chunk needs; the second pass has the region tiers'
			// run time to finish them in.
			if (budget && ++trips > budget) L.overflow = 1;
			if (L.overflow) L.state = SD_DONE;
		}
	}
	// work counters for the algorithmic-bytes model: slow path = two 64-B blocks, fast = one
	for (int off = 32; off > 0; off >>= 1) { tot_slow += __shfl_down(tot_slow, off); tot_fast += __shfl_down(tot_fast, off); tot_over += __shfl_down(tot_over, off); }
	if (prof) {
		long long pub = pc_pub;
		for (int off = 32; off > 0; off >>= 1) { long long o = __shfl_down(pub, off); pub = pub > o ? pub : o; }
		if ((threadIdx.x & 63) == 0) {
			atomicAdd(&counters[48], (unsigned long long)(clock64() - pc_t0)); atomicAdd(&counters[49], (unsigned long long)pc_cold);
			atomicAdd(&counters[50], (unsigned long long)pub); atomicAdd(&counters[51], (unsigned long long)trip); atomicAdd(&counters[52], (unsigned long long)pc_cold_n);
		}
	}
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&counters[0], 2ull * tot_slow); atomicAdd(&counters[1], (unsigned long long)tot_fast);
		if (tot_over) atomicAdd(&counters[119], (unsigned long long)tot_over);   // strand searches left with a negative count (k_seedt.hip)
		__threadfence();
		atomicExch(&slab_busy[slab], 0u);
	}
}

// K3: bwt_sa (lib/aln/bwt.c:87-97) -- LF-walk to the next sampled rank; one lane per lookup.
__global__ void __launch_bounds__(256)
k_sa(DevIndex ix, const bsx_sa_job_t *jobs, long long n, uint64_t *pos, unsigned long long *counters)
{
	long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t steps = 0, calls = 0;
	for (long long i = gid; i < n; i += (long long)gridDim.x * blockDim.x) {
		const DevFmi f = dev_fmi_pick(ix, jobs[i].parent);
		uint64_t k = jobs[i].k, sa = 0;
		while (k & f.sa_mask) {
			// bwt_invPsi (bwt.c:54-60): symbol at k and its rank come from the same 64-byte block
			if (k == f.primary) { k = 0; ++sa; ++steps; continue; }
			uint64_t x = k - (k > f.primary);
			const DevBlock B = dev_load_block4(f.bwt, x);
			const int c = dev_planes_symbol(B, (int)(x & 127));
			uint32_t c4[4];
			dev_planes_count4(B, (int)(x & 127), c4[0], c4[1], c4[2], c4[3]);
			const uint64_t b0 = (uint64_t)B.v0.y << 32 | B.v0.x, b1 = (uint64_t)B.v0.w << 32 | B.v0.z, b2 = (uint64_t)B.v1.y << 32 | B.v1.x, b3 = (uint64_t)B.v1.w << 32 | B.v1.z;
			k = dev_L2(f, c) + (c == 0 ? b0 + c4[0] : c == 1 ? b1 + c4[1] : c == 2 ? b2 + c4[2] : b3 + c4[3]);
			++sa; ++steps;
		}
		pos[i] = sa + f.sa[k >> f.sa_shift];
		++calls;
	}
	for (int off = 32; off > 0; off >>= 1) { steps += __shfl_down(steps, off); calls += __shfl_down(calls, off); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[2], (unsigned long long)steps); atomicAdd(&counters[3], (unsigned long long)calls); }
}

void launch_seed(hipStream_t st, int grid, const DevIndex &ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, const SeedParams &P,
                 DevIntv *scratch, int list_cap, int mem_cap, DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
                 long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
                 int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof, uint32_t *qpack, unsigned long long direct_off)
{
	// 165 VGPRs and 11 KB of LDS per wave: three waves per SIMD
	// the full state machine runs every (cold_mask + 1)-th trip, or when more than cold_lanes lanes wait for it (measured in round 3,
	// the launch does not care: 208.2-209.8 ms from every 2nd trip / 16 lanes to every 16th / 32): what
	// bounds it is the vector issue of the trips themselves -- 820 wave64 instructions at four cycles each on a 16-wide SIMD, three waves deep
	const unsigned int cold_mask = 4u - 1u;
	const int cold_lanes = 24;
	// the table form (k_seedt.hip) whenever the index has its table; $BSX_SEED_FORM=classic (or a number: the forms below) keeps this kernel
	const bool classic = bsx_tune_is_set("seed_form");   // (a setting of the library: the tests and bench.py switch it between calls)
	if (!classic) {
		launch_seedt(st, grid, ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap, out, out_cap, out_cursor, task_off, task_n, task_cursor, counters,
		             quota, slab_busy, n_slabs, trip_budget, prof, qpack, direct_off);
		return;
	}
	hipLaunchKernelGGL((k_seed<3, 24>), dim3(grid * (4 / SEED_WPB)), dim3(64 * SEED_WPB), 0, st, /* `grid` counts groups of four waves */ ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap, out, out_cap, out_cursor,
	                   task_off, task_n, task_cursor, counters, quota, slab_busy, n_slabs, trip_budget, prof, cold_mask, cold_lanes);
}
// the file's 2-bit symbol fields of every block into the device's bit planes (dev_common.hpp), in place; a thread per block
__global__ void __launch_bounds__(256)
k_bwt_planes(uint32_t *bwt, unsigned long long n_blocks)
{
	const unsigned long long blk = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (blk >= n_blocks) return;
	uint4 *p = reinterpret_cast<uint4*>(bwt + blk * 16);
	const uint4 v2 = p[2], v3 = p[3];
	const uint32_t w[8] = {v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
	uint32_t lo[4], hi[4];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		uint32_t pl[2], ph[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) { // the even (low) and odd (high) bits of a word squeezed into 16 bits each, order kept
			uint32_t e = w[2 * q + h] & 0x55555555u, o = (w[2 * q + h] >> 1) & 0x55555555u;
			e = (e | e >> 1) & 0x33333333u; e = (e | e >> 2) & 0x0f0f0f0fu; e = (e | e >> 4) & 0x00ff00ffu; e = (e | e >> 8) & 0x0000ffffu;
			o = (o | o >> 1) & 0x33333333u; o = (o | o >> 2) & 0x0f0f0f0fu; o = (o | o >> 4) & 0x00ff00ffu; o = (o | o >> 8) & 0x0000ffffu;
			pl[h] = e; ph[h] = o;
		}
		lo[q] = pl[0] << 16 | pl[1]; hi[q] = ph[0] << 16 | ph[1];
	}
	p[2] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
	p[3] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
}
void launch_bwt_planes(hipStream_t st, uint32_t *bwt, unsigned long long n_words)
{
	const unsigned long long n_blocks = (n_words + 15) / 16;
	hipLaunchKernelGGL(k_bwt_planes, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, st, bwt, n_blocks);
}

void launch_sa(hipStream_t st, int grid, const DevIndex &ix, const bsx_sa_job_t *jobs, long long n, uint64_t *pos, unsigned long long *counters)
{
	hipLaunchKernelGGL(k_sa, dim3(grid), dim3(256), 0, st, ix, jobs, n, pos, counters);
}
