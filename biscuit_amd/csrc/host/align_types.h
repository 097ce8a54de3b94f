/* align_types.h -- host-side records of the alignment path (seeds, chains, regions) and the
 * per-stage entry points.  Field meaning follows the reference structs cited at each type. */
#ifndef BSX_ALIGN_TYPES_H
#define BSX_ALIGN_TYPES_H

#include "bsx_core.h"

/* mem_seed_t (lib/aln/memchain.h:70-74) */
typedef struct {
	int64_t rbeg;
	int32_t qbeg, len;
	int32_t score;
} seed_t;
typedef BSX_VEC(seed_t) seed_v;

/* mem_chain_t (lib/aln/memchain.h:84-94) */
typedef struct {
	int first;
	int rid;
	uint32_t w;         /* 29-bit weight in the reference */
	int kept;
	int is_alt;
	float frac_rep;
	int64_t pos;
	seed_v seeds, seeds_extra;
} chain_t;
typedef BSX_VEC(chain_t) chain_v;

/* mem_alnreg_t (lib/aln/mem_alnreg.h:34-66) */
typedef struct {
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp;
	int is_alt;
	float frac_rep;
	uint64_t hash;
	uint8_t bss, parent, read_in_pair;
	/* SAM side */
	int pos, flag, NM, n_cigar;
	uint32_t is_rev, sam_set;
	unsigned mapq;
	uint32_t ZC, ZR;
	int bss_u;
	uint32_t *cigar;     /* n_cigar ops followed by the NUL-terminated MD string */
} reg_t;
typedef BSX_VEC(reg_t) reg_vv;
typedef struct { size_t n, m; reg_t *a; size_t n_pri; } reg_v;

/* ---------------- chain.c ---------------- */
/* mem_chain (memchain.c:268-393) on pre-computed SA positions.
 * intv[0..n_intv): sorted intervals of this strand search; pos_off[i]..pos_off[i+1]: positions of
 * occurrences 0.. of interval i (bwt_sa(x0+k)).  Returns 0, or 1+i when interval i needs more
 * occurrences than were supplied (the caller looks more up and calls again). */
int bsx_chain_build(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int l_seq, int parent,
                    const bsx_intv_t *intv, int n_intv, const uint64_t *pos, const int64_t *pos_off,
                    bsx_btree_t *tree, chain_v *chains);
void bsx_chain_filter(const bsx_opt_t *opt, chain_v *chains);                 /* mem_chain_flt, memchain.c:406-488 */
void bsx_chain_free(chain_v *chains);
int  bsx_cal_max_gap(const bsx_opt_t *opt, int qlen);                          /* memchain.c:576-582 */
void bsx_chain_ref_span(const bsx_opt_t *opt, int l_query, int64_t l_pac, const chain_t *c, int64_t rmax[2]); /* memchain.c:585-605 */

/* ---------------- region.c ---------------- */
typedef int (*bsx_glb_score_fn)(void *ud, const reg_t *a, const reg_t *b, int w, int *score); /* returns 0 ok, 1 = not available yet */
void bsx_regs_sort_dedup(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int can_merge, reg_v *regs,
                         bsx_glb_score_fn score_fn, void *ud, int *missing);   /* mem_sort_deduplicate, mem_alnreg.c:112-202 */
/* a rescued hit into the mate's list by score, then mem_sort_deduplicate without merging (mem_alnreg.c:478-488), hit after hit onto the
 * same list: st (zeroed before the first hit, freed after the last) carries the list's order by end between the calls */
typedef struct { int valid, m; int *ord; } bsx_regs_inc_t;
void bsx_regs_insert_dedup(const bsx_opt_t *opt, const bsx_refmeta_t *ref, reg_v *regs, const reg_t *b, bsx_regs_inc_t *st, bsx_glb_score_fn no_score_fn);
void bsx_regs_inc_free(bsx_regs_inc_t *st);
void bsx_mark_primary(const bsx_opt_t *opt, reg_v *regs, int64_t id);          /* mem_mark_primary_se, mem_alnreg.c:290-380 */
bsx_pestat_t bsx_pestat(const bsx_opt_t *opt, const bsx_refmeta_t *ref, int n, const reg_v *regs); /* mem_pestat, mem_pair.c:60-144 */
void bsx_pair(const bsx_opt_t *opt, const bsx_refmeta_t *ref, const bsx_pestat_t *pes, reg_v pair[2], int id,
              int *score, int *sub, int *n_sub, int z[2]);                     /* mem_pair, mem_pair.c:147-270 */
int  bsx_reg_isize(const bsx_refmeta_t *ref, const reg_t *r1, const reg_t *r2, int64_t *isize); /* mem_alnreg_isize */
int  bsx_approx_mapq_se(const bsx_opt_t *opt, const reg_t *a);                 /* mem_approx_mapq_se, bwamem.c:134-157 */

#endif
