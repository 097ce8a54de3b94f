/* opts.c -- alignment options and scoring matrices.
 * Defaults: mem_opt_init (lib/aln/bwamem.c:77-128); matrices: bwa_fill_scmat{,_ct,_ga}
 * (lib/aln/bwa.c:146-182), indexed mat[ref*5+read]. */
#include <math.h>
#include "bsx_core.h"

static void fill_mat(int a, int b, int8_t mat[25], int ri, int qj)
{
	/* (ri,qj) = the one off-diagonal (reference base, read base) pair that still scores +a:
	 * C/T for the C>T matrix, G/A for the G>A matrix, none (-1,-1) for the plain one */
	int i, j, k = 0;
	for (i = 0; i < 4; ++i) {
		for (j = 0; j < 4; ++j) mat[k++] = (int8_t)((i == j || (i == ri && j == qj)) ? a : -b);
		mat[k++] = -1;
	}
	for (j = 0; j < 5; ++j) mat[k++] = -1;
}

BSX_API void bsx_opt_fill_matrices(bsx_opt_t *o)
{
	fill_mat(o->a, o->b, o->mat, -1, -1);
	fill_mat(o->a, o->b, o->ctmat, 1, 3);
	fill_mat(o->a, o->b, o->gamat, 2, 0);
}

BSX_API void bsx_opt_init(bsx_opt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->a = 1; o->b = 2;
	o->o_del = o->o_ins = 6;
	o->e_del = o->e_ins = 1;
	o->w = 100;
	o->T = 30;
	o->zdrop = 100;
	o->pen_unpaired = 17;
	o->pen_clip5 = o->pen_clip3 = 10;
	o->max_mem_intv = 20;
	o->min_seed_len = 19;
	o->split_width = 10;
	o->max_occ = 500;
	o->max_chain_gap = 10000;
	o->max_ins = 5000;
	o->mask_level = 0.50;
	o->drop_ratio = 0.50;
	o->XA_drop_ratio = 0.80;
	o->split_factor = 1.5;
	o->chunk_size = 10000000;
	o->n_threads = 1;
	o->max_XA_hits = 5;
	o->max_XA_hits_alt = 5;
	o->max_matesw = 50;
	o->mask_level_redun = 0.95;
	o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30;
	o->mapQ_coef_len = 50;
	o->mapQ_coef_fac = log(o->mapQ_coef_len); /* int field: truncates 3.912 to 3 (bwamem.h:81) */
	bsx_opt_fill_matrices(o);
}
