/* pipeline.h -- per-chunk working records of the batch pipeline (pipeline.c) */
#ifndef BSX_PIPELINE_H
#define BSX_PIPELINE_H

#include "align_types.h"

/* one strand search (mem_align1_core call, lib/aln/bwamem.c:183-208) being turned into regions */
typedef struct {
	/* header: every strand search */
	int read_idx, parent;
	uint32_t qoff;            /* of the clipped read in the chunk read buffer */
	int l_query;
	const uint8_t *query;     /* clipped raw read on the host */
	chain_v chains;
	BSX_VEC(reg_t) regs;
	int has_job, done;
	/* state machine (extend.c): only touched for the strand searches the host chains itself (bsx_c2r_init) */
	int ci, chain_open, pass, n0;
	uint64_t *srt; int n_srt, m_srt, k;
	int stage, tryi;
	int64_t rmax[2]; int rid;
	reg_t cur; int aw[2]; int sc0;
	bsx_ext_job_t job;
} c2r_t;
#define C2R_HEADER_BYTES offsetof(c2r_t, ci)
static inline void bsx_c2r_init(c2r_t *t) { memset((char*)t + C2R_HEADER_BYTES, 0, sizeof(c2r_t) - C2R_HEADER_BYTES); }

int  bsx_c2r_advance(const bsx_opt_t *opt, const bsx_index_t *idx, c2r_t *t);
void bsx_c2r_consume(const bsx_opt_t *opt, const bsx_index_t *idx, c2r_t *t, const bsx_ext_res_t *res);
void bsx_c2r_release(c2r_t *t);

/* ---------------- sam.c ---------------- */
/* one computed CIGAR (what mem_alnreg_setSAM leaves in a region, lib/aln/mem_alnreg_format.c:40-123) */
typedef struct {
	int valid;
	int pos, n_cigar, NM, bss_u;
	uint32_t is_rev, ZC, ZR;
	uint32_t *cigar;          /* ops + MD text, malloc'd */
} samrec_t;

typedef struct {
	/* plan mode: collect the regions whose CIGAR will be needed instead of formatting */
	int plan;
	BSX_VEC(int) want[2];     /* region indices per read of the pair (or [0] for SE) */
	samrec_t *table[2];       /* per region, filled between plan and final pass */
	/* tests: with trace != NULL the planning pass notes every record it would write: read of the pair, index of the region in its
	 * list (-1: the unmapped stand-in), index of the mate's region (-1: unmapped stand-in, -2: none), flag and mapq of the region
	 * at that moment, is_primary */
	int (*trace)[6]; int n_trace, m_trace;
} samctx_t;

void bsx_reg2sam_se(const bsx_opt_t *opt, const bsx_index_t *idx, bsx_read_t *s, reg_v *regs, samctx_t *ctx, const char *rg_id);
void bsx_reg2sam_pe(const bsx_opt_t *opt, const bsx_index_t *idx, uint64_t id, bsx_read_t s[2], reg_v regs[2],
                    const bsx_pestat_t *pes, samctx_t *ctx, const char *rg_id);
/* K6 job for one region (band inference of mem_alnreg_setSAM) */
void bsx_setsam_job(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_read_t *s, uint32_t qoff, const reg_t *reg, bsx_glb_job_t *job);
/* finish a region's SAM record from the device CIGAR: MD/NM/ZC/ZR, D-squeezing, clipping, position */
void bsx_setsam_finish(const bsx_opt_t *opt, const bsx_index_t *idx, const bsx_read_t *s, const reg_t *reg,
                       const uint32_t *cg, int n_cigar, samrec_t *out, const bsx_glb_tag_t *tag, const char *tag_md);

extern char bsx_rg_id[256];

#endif
