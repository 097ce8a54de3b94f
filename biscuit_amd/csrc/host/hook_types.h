/* hook_types.h -- the flat region record the test hooks exchange with Python (hooks.c, pipeline.c) */
#ifndef BSX_HOOK_TYPES_H
#define BSX_HOOK_TYPES_H
#include <string.h>
#include "align_types.h"

typedef struct {
	int64_t rb, re;
	int32_t qb, qe, rid, score, truesc, sub, alt_sc, csub, sub_n, w, seedcov, secondary, secondary_all, seedlen0, n_comp, is_alt;
	uint64_t hash;
	int32_t flag, mapq;
	float frac_rep;
	uint8_t bss, parent, pad[2];
	/* SAM side (mem_alnreg_setSAM's outputs) */
	int32_t pos, n_cigar, NM, bss_u;
	uint32_t is_rev, ZC, ZR, pad2;
	uint32_t *cigar;          /* n_cigar operations followed by the NUL-terminated MD string, or NULL */
} bsx_hook_reg_t;

static inline void bsx_hook_to_reg(const bsx_hook_reg_t *h, reg_t *r)
{
	memset(r, 0, sizeof(*r));
	r->rb = h->rb; r->re = h->re; r->qb = h->qb; r->qe = h->qe; r->rid = h->rid; r->score = h->score; r->truesc = h->truesc; r->sub = h->sub;
	r->alt_sc = h->alt_sc; r->csub = h->csub; r->sub_n = h->sub_n; r->w = h->w; r->seedcov = h->seedcov; r->secondary = h->secondary;
	r->secondary_all = h->secondary_all; r->seedlen0 = h->seedlen0; r->n_comp = h->n_comp; r->is_alt = h->is_alt; r->hash = h->hash;
	r->bss = h->bss; r->parent = h->parent; r->flag = h->flag; r->mapq = (unsigned)h->mapq; r->frac_rep = h->frac_rep;
	r->pos = h->pos; r->n_cigar = h->n_cigar; r->NM = h->NM; r->bss_u = h->bss_u; r->is_rev = h->is_rev; r->ZC = h->ZC; r->ZR = h->ZR; r->cigar = h->cigar;
}
static inline void bsx_hook_from_reg(const reg_t *r, bsx_hook_reg_t *h)
{
	memset(h, 0, sizeof(*h));
	h->rb = r->rb; h->re = r->re; h->qb = r->qb; h->qe = r->qe; h->rid = r->rid; h->score = r->score; h->truesc = r->truesc; h->sub = r->sub;
	h->alt_sc = r->alt_sc; h->csub = r->csub; h->sub_n = r->sub_n; h->w = r->w; h->seedcov = r->seedcov; h->secondary = r->secondary;
	h->secondary_all = r->secondary_all; h->seedlen0 = r->seedlen0; h->n_comp = r->n_comp; h->is_alt = r->is_alt; h->hash = r->hash;
	h->bss = r->bss; h->parent = r->parent; h->flag = r->flag; h->mapq = (int32_t)r->mapq; h->frac_rep = r->frac_rep;
	h->pos = r->pos; h->n_cigar = r->n_cigar; h->NM = r->NM; h->bss_u = r->bss_u; h->is_rev = r->is_rev; h->ZC = r->ZC; h->ZR = r->ZR; h->cigar = r->cigar;
}
#endif
