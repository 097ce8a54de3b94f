// k_seed.hip -- K1+K2 (FM-index SMEM seeding) and K3 (suffix-array lookup) kernels, gfx950.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "seed_core.hpp"
#include "kernels.h"

#define SEED_LDS_WORDS 32   // 256 bases per lane in LDS: 8 KB per wave

// One lane = one strand search at a time; lanes pull the next task from a global cursor as soon
// as they finish (reads differ a lot in seeding work), so a wave stays full until the queue drains.
// Every trip of the outer loop issues the FM-block gathers of all 64 lanes together.
// A lane retires after `quota` tasks (0 = never): workgroups then have a bounded life and the launch is
// many more workgroups than fit on the chip, which lets kernels of a higher-priority stream (the back half of
// the previous chunk) get compute units while this one is running.  Scratch slabs are therefore not tied to
// the workgroup index: each wave takes a free slab and gives it back when it exits.
template <int OCC>
__global__ void __launch_bounds__(256, OCC)
k_seed(DevIndex ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, SeedParams P,
       DevIntv *scratch, int list_cap, int mem_cap,
       DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
       long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
       int quota, unsigned int *slab_busy, int n_slabs, int trip_budget)
{
	// per-wave slab, lane-interleaved: entry i of lane l sits at slab[i*64 + l], so the 64 lanes'
	// accesses to the same list position form one contiguous 2 KB run (coalesced, one TLB page)
	int slab = 0;
	if ((threadIdx.x & 63) == 0) {
		unsigned int h = (unsigned int)((blockIdx.x * 4u + (threadIdx.x >> 6)) % (unsigned int)n_slabs);
		while (atomicCAS(&slab_busy[h], 0u, 1u) != 0u) h = h + 1 == (unsigned int)n_slabs ? 0 : h + 1;
		slab = (int)h;
	}
	slab = __shfl(slab, 0);
	const size_t wave_id = (size_t)slab;
	const size_t per_lane = (size_t)2 * list_cap + mem_cap;
	SeedLane L;
	L.stride = 64;
	L.bufA = scratch + wave_id * per_lane * 64 + (threadIdx.x & 63);
	L.bufB = L.bufA + (size_t)list_cap * 64;
	L.mem = L.bufB + (size_t)list_cap * 64;
	L.list_cap = list_cap; L.mem_cap = mem_cap;
	// the read, bisulfite-converted and packed 8 bases/word, lane-interleaved in LDS (<= 256 bases)
	__shared__ uint32_t s_read[4][SEED_LDS_WORDS][64];
	uint32_t *my_read = &s_read[threadIdx.x >> 6][0][threadIdx.x & 63];
	L.qlds = nullptr;
	L.state = SD_DONE;
	L.n_slow = L.n_fast = 0;
	int task = -1, retired = 0, taken = 0, trips = 0, budget = 0;
	uint32_t tot_slow = 0, tot_fast = 0;

	unsigned int trip = 0;
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
		if (cold && ((trip & 3u) == 0 || __popcll(cm) > 24 || __ballot(need) == 0)) {
			for (;;) {
				if (L.state == SD_DONE) {
					if (task >= 0) { // publish the finished task
						unsigned long long base = 0;
						int n = L.mem_n;
						if (n > 0 && !L.overflow) {
							base = atomicAdd(out_cursor, (unsigned long long)n);
							if (base + n <= out_cap) for (int k = 0; k < n; ++k) out[base + k] = L.mem[(size_t)k * 64];
							else L.overflow = 1;
						}
						task_off[task] = (long long)base;
						task_n[task] = L.overflow ? -n - 1 : n;   // any negative count: seed this strand search again
						tot_slow += L.n_slow; tot_fast += L.n_fast;
						task = -1;
					}
					if (quota && taken >= quota) { retired = 1; break; }
					unsigned int t = atomicAdd(task_cursor, 1u);
					if (t >= (unsigned int)n_tasks) { retired = 1; break; }
					++taken;
					task = (int)t;
					L.q = reads + tasks[t].qoff; L.len = tasks[t].len; L.parent = tasks[t].parent;
					L.qlds = nullptr;
					if (L.len <= SEED_LDS_WORDS * 8) {
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
		if (__all(retired)) break;
		if (need) {
			const int which = L.ext_which ? !L.parent : L.parent;
			DevFmi fx; fx.primary = dev_ix_primary(ix, which); fx.bwt = dev_ix_bwt(ix, which);
			fx.L2[0] = fx.L2[1] = fx.L2[2] = fx.L2[3] = fx.L2[4] = dev_ix_L2(ix, which, L.ext_c);
			DevIntv ok = dev_extend(fx, L.ext_in, L.ext_back, L.ext_c, L.n_slow, L.n_fast);
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
	for (int off = 32; off > 0; off >>= 1) { tot_slow += __shfl_down(tot_slow, off); tot_fast += __shfl_down(tot_fast, off); }
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(&counters[0], 2ull * tot_slow); atomicAdd(&counters[1], (unsigned long long)tot_fast);
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
			uint64_t base[4]; uint32_t w[8], c4[4];
			dev_load_block(f.bwt, x, base, w);
			int c = (w[(x & 127) >> 4] >> ((~x & 15) << 1)) & 3;
			dev_block_count(w, (int)(x & 127), c4);
			k = dev_L2(f, c) + (c == 0 ? base[0] + c4[0] : c == 1 ? base[1] + c4[1] : c == 2 ? base[2] + c4[2] : base[3] + c4[3]);
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
                 int quota, unsigned int *slab_busy, int n_slabs, int trip_budget)
{
	static const int occ = getenv("BSX_SEED_OCC") ? atoi(getenv("BSX_SEED_OCC")) : 3;   // waves per SIMD the register allocation targets
	if (occ <= 3)
		hipLaunchKernelGGL(k_seed<3>, dim3(grid), dim3(256), 0, st, ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap, out, out_cap, out_cursor,
		                   task_off, task_n, task_cursor, counters, quota, slab_busy, n_slabs, trip_budget);
	else if (occ == 4)
		hipLaunchKernelGGL(k_seed<4>, dim3(grid), dim3(256), 0, st, ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap, out, out_cap, out_cursor,
		                   task_off, task_n, task_cursor, counters, quota, slab_busy, n_slabs, trip_budget);
	else
		hipLaunchKernelGGL(k_seed<5>, dim3(grid), dim3(256), 0, st, ix, reads, tasks, n_tasks, P, scratch, list_cap, mem_cap, out, out_cap, out_cursor,
		                   task_off, task_n, task_cursor, counters, quota, slab_busy, n_slabs, trip_budget);
}
void launch_sa(hipStream_t st, int grid, const DevIndex &ix, const bsx_sa_job_t *jobs, long long n, uint64_t *pos, unsigned long long *counters)
{
	hipLaunchKernelGGL(k_sa, dim3(grid), dim3(256), 0, st, ix, jobs, n, pos, counters);
}
