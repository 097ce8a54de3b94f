// kernels.h -- launchers of the gfx950 kernels (defined in k_*.hip), used by shim.hip
#pragma once
#include <hip/hip_runtime.h>
#include "dev_common.hpp"
#include "seed_core.hpp"
#include "seed_tab.hpp"

void launch_seed(hipStream_t st, int grid, const DevIndex &ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, const SeedParams &P,
                 DevIntv *scratch, int list_cap, int mem_cap, DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
                 long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
                 int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof = 0, uint32_t *qpack = nullptr, unsigned long long direct_off = ~0ull);   // direct_off (table form only): the lists of strand search t go to out[direct_off + t * mem_cap ...] and stay there (task_off says so; no cursor); scratch holds n_slabs per-wave slabs; slab_busy[n_slabs] zeroed once; qpack: seedt_pack_bytes(n_tasks) of scratch for the table form (the reads as base-3 digits)
// the same over the table of k-mer intervals (k_seedt.hip, seed_tab.hpp): what launch_seed runs when ix.tab.K >= 2 (and $BSX_SEED_FORM is not "classic");
// counters[120] += table entries read
void launch_seedt(hipStream_t st, int grid, const DevIndex &ix, const uint8_t *reads, const bsx_seed_task_t *tasks, int n_tasks, const SeedParams &P,
                  DevIntv *scratch, int list_cap, int mem_cap, DevIntv *out, unsigned long long out_cap, unsigned long long *out_cursor,
                  long long *task_off, int *task_n, unsigned int *task_cursor, unsigned long long *counters,
                  int quota, unsigned int *slab_busy, int n_slabs, int trip_budget, int prof = 0, uint32_t *qpack = nullptr, unsigned long long direct_off = ~0ull);
size_t seedt_pack_bytes(long long n_tasks);
void launch_gather_lists(hipStream_t st, const DevIntv *src, const long long *off, const int *cnt, const long long *which, const long long *dst_off, long long n_list, DevIntv *dst);
// the table of one converted index, K levels (seed_tab_entries(K) entries of 16 bytes at `table`), from the index resident in ix
int seedtab_build(hipStream_t st, const DevIndex &ix, int parent, int K, void *table);
// the symbols of every 64-byte block from the file's 2-bit fields into the device's two bit planes, in place (n_words: the .bwt body)
void launch_bwt_planes(hipStream_t st, uint32_t *bwt, unsigned long long n_words);
void launch_sa(hipStream_t st, int grid, const DevIndex &ix, const bsx_sa_job_t *jobs, long long n, uint64_t *pos, unsigned long long *counters);
// DP kernels: one wavefront per job
void launch_extend(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order,
                   long long n, bsx_ext_res_t *res, int qcap, int nc, int n_cu);
// the same for queries whose rows outgrow LDS (longer than 16 k bases): rows in `rows`, extend_hbm_row_bytes(qcap) per wave, blocks x 4 waves
size_t extend_hbm_row_bytes(int qcap);
void launch_extend_hbm(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, const int *order,
                       long long n, bsx_ext_res_t *res, int qcap, int blocks, void *rows);
size_t global_hbm_row_bytes(int qcap);
void launch_global_hbm(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order,
                       long long n, bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap, int blocks,
                       bsx_glb_tag_t *tags, char *md_pool, unsigned long long md_cap, unsigned long long *md_cursor, void *rows);   // one wave per workgroup
void launch_sw(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order,
               long long n, bsx_sw_res_t *res, unsigned long long *bscratch, int bcap, int blocks, int nc);
void launch_swl(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_sw_job_t *jobs, const int *order,
                long long n, bsx_sw_res_t *res, unsigned long long *bscratch, int bcap, int blocks, int slen_max);   // k_swl.hip: byte-sized jobs, four to a wavefront
void launch_global(hipStream_t st, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_glb_job_t *jobs, const int *order,
                   long long n, bsx_glb_res_t *res, uint32_t *pool, uint8_t *zscratch, size_t zstride, int qcap, int nc, int blocks, int wpb,
                   bsx_glb_tag_t *tags = nullptr, char *md_pool = nullptr, unsigned long long md_cap = 0, unsigned long long *md_cursor = nullptr, int tcap = 0);
// K4 four to a wavefront (k_ext4.hip): a row of 16 lanes per job, persistent rows taking jobs off *cursor (zero at launch).
// launch_x4: the extensions of the best seed of every chain the tiers exported (records with has_ext), written into the records ahead of
// launch_c2r; jobs = room for job_cap jobs of x4_job_bytes(), ctr32[0..3]: job counts and cursor (zeroed by the call).
// launch_ext4_batch: plain ksw_extend2 jobs through the same rows (tests); jobs it cannot hold (query longer than x4_max_query(16),
// scores of 2^21 or more) are answered with score = X4_DECLINED.
#define X4_DECLINED (-0x7fffffff)
// tests: plain jobs through ext_dp_win (ext_dp.hpp: rows in a register window that follows the band, queries of any length), a wavefront per job
void launch_extwin_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res, long long n);
int x4_max_query(int ncq);
size_t x4_job_bytes(void);
struct RgXPoolArg;
void launch_x4(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
               long long n_tasks, const RgXPoolArg &X, void *jobs, unsigned long long job_cap, unsigned int *ctr32, unsigned long long *prof);
// K4 a lane per job (k_extl.hip) for the narrow queue of launch_x4's pool; called by launch_x4
void launch_extl(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, void *jobs, unsigned int jcap,
                 unsigned int *ctr32, unsigned char *xbase, long long n_upper, unsigned long long *prof);
void launch_extl_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                       unsigned int n, unsigned int *ctr32);
void launch_ext4_batch(hipStream_t st, int n_cu, const DevIndex &ix, const DevScoring &sc, const uint8_t *reads, const bsx_ext_job_t *jobs, bsx_ext_res_t *res,
                       unsigned int n, unsigned int *cursor, int max_qlen);
// K3+C1+C2+C4 fused: one wavefront per strand search, from the dense interval lists of launch_seed to alignment regions.
// Tier 1 keeps its tables in LDS; tiers 2 and 3 run what did not fit over per-wave slabs in HBM (grid * 4 slabs of
// regions_slab_bytes(tier)); a tier appends what it declines for table size to the next tier's list.
size_t regions_slab_bytes(int tier);
// where the LDS tiers leave the chains that survive the filter for launch_c2r (chains -> regions): a byte pool with a bump cursor,
// the block of task t at xoff[t], and the list of tasks that have one
struct RgXPoolArg { unsigned char *base; unsigned long long cap; unsigned long long *cursor; long long *xoff; int *xlist; unsigned int *xcount;
                    int ext; };   // ext: the records leave room for the extensions launch_x4 makes ahead of launch_c2r
void launch_regions(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                    const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                    bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                    unsigned int *task_cursor, int *retry_list, unsigned int *retry_count, int quota, unsigned long long *counters,
                    const long long *pos_off, const unsigned long long *pos, const unsigned char *cls, const RgXPoolArg &X, int long_reads = 0);   // cls[t] != 0: not for this tier (launch_occ)
// the tier in between: LDS tables four times the first tier's; consumes the first tier's list, appends to the second tier's
void launch_regions_mid(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                        const bsx_seed_task_t *tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                        bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                        const int *list, const unsigned int *count, unsigned int *cursor, int *next_list, unsigned int *next_count,
                        unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, const RgXPoolArg &X, int quota, int long_reads = 0);   // quota: strand searches per wave; long_reads: the instantiation for reads up to regions_long_max_query()
// chains -> regions for everything the two launches above exported; what does not fit its tables goes on next_list
void launch_c2r(hipStream_t st, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
                const RgXPoolArg &X, bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                unsigned int *cursor, int *next_list, unsigned int *next_count, unsigned long long *counters, int quota, int long_reads = 0, void *slab = nullptr);   // next_list may be null (long reads: what outgrows the tables is left to the caller)   // long_reads 2: larger LDS tables (what the first launch declines); 3: the large tables in HBM, `slab` = grid x c2r_hbm_slab_bytes()
size_t c2r_hbm_slab_bytes(void);
// C3 between the tiers and launch_c2r: the seed-SW filter of the exported strand searches it applies to (mem_flt_chained_seeds, memchain.c:537-568)
void launch_seedsw(hipStream_t st, int grid, int n_cu, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads, const bsx_seed_task_t *tasks,
                   const RgXPoolArg &XA, unsigned int *cursor, unsigned int *count_cursor, void *jobs, unsigned int job_cap, unsigned long long *counters);
size_t seedsw_job_bytes(void);
int regions_long_max_query(void);
void launch_regions_slab(hipStream_t st, int tier, int grid, const DevIndex &ix, const DevScoring &sc, const RegParams &P, const uint8_t *reads,
                         const bsx_seed_task_t *tasks, const DevIntv *seeds_dense, const long long *task_off, const int *task_n,
                         bsx_region_t *out, unsigned long long out_cap, unsigned long long *out_cursor, long long *reg_off, int *reg_n,
                         const int *list, const unsigned int *count, unsigned int *cursor, void *slabs, int *next_list, unsigned int *next_count,
                         unsigned long long *counters, const long long *pos_off, const unsigned long long *pos, const RgXPoolArg *X = nullptr);   // X: export the chains (as the LDS tiers do) instead of making the regions
// a tier's list put in order of decreasing size (occurrences to visit), longest strand search first: what a launch that lasts as long as its longest one wants
void launch_order_list(hipStream_t st, int *list, const unsigned int *count, const DevIntv *seeds_dense, const long long *task_off, const int *task_n, int max_occ);
// K3 ahead of the region kernels: SA ranks of all occurrences of all strand searches listed into desc (pos_off[t] = where task
// t's start, -1 = none listed), then turned into reference positions in place.  pos_off/pos feed launch_regions*.
void launch_occ(hipStream_t st, int n_cu, const DevIndex &ix, const bsx_seed_task_t *tasks, int n_tasks, const DevIntv *seeds_dense, const long long *task_off,
                const int *task_n, int max_occ, unsigned long long *desc, unsigned long long desc_cap, unsigned long long *cursor, long long *pos_off,
                unsigned long long *counters, unsigned char *cls, unsigned long long *start = nullptr, int *early_list = nullptr, unsigned int *early_count = nullptr);   // early_list: the strand searches only the last HBM tier's tables hold are listed there (cls 3; the first tier then skips them)   // start: an 8-byte device slot; set = only the ranks this call adds to the pool are walked   // cls[t]: the first tier whose interval/occurrence tables hold task t
// out[j] = SA[j * intv] for j < n, from the (sparser) samples ix currently holds: the denser suffix-array sample kept in HBM
void launch_sa_dense(hipStream_t st, int n_cu, const DevIndex &ix, int parent, unsigned int intv, unsigned long long n, unsigned long long *out);
// C5 (k_dedup.hip): mem_sort_deduplicate of every read over the regions of the chunk, a lane per read
int dedup_cap(void);
void launch_dedup(hipStream_t st, const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
                  long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, int *out_n, unsigned char *out_idx,
                  int *long_list = nullptr, unsigned int *long_count = nullptr);   // long_list (2 x n_reads ints) / long_count (2 counters, zeroed): the reads k_dedup_long takes
// ... and a wavefront per read for the reads with more regions than that (up to dedup_long_cap()): 16-bit indices, one list behind the other in pool
int dedup_long_cap(void);
void launch_dedup_long(hipStream_t st, int n_cu, const bsx_region_t *regs, const long long *reg_off, const int *reg_n, int n_reads, int per_read,
                       long long l_pac, int max_chain_gap, int opt_w, float mask_level_redun, const int *long_list, const unsigned int *long_count,
                       int *out_n, long long *out_off, unsigned short *pool, unsigned long long pool_cap, unsigned long long *pool_cursor);

// mate rescue's plan on the device (k_msw.hip): candidates -> jobs (binned for k_swl) + a record per pair
size_t msw_pair_bytes(void);
int msw_hist_bins(void);
void launch_msw_plan(hipStream_t st, int n_cu, const DevIndex &ix, const bsx_region_t *regs, const long long *r_off, const int *r_n,
                     const int *dd_n, const unsigned char *dd_idx, const long long *dd_off, const unsigned short *dd_pool, int dd_cap, int per_read,
                     const unsigned int *roff, int low, int high, int pen_unpaired, int max_matesw, int min_seed_len, int a, int p0, int n_pairs,
                     bsx_sw_job_t *jobs, unsigned int job_cap, unsigned int *job_count, unsigned int *hist, void *table, int *order);
