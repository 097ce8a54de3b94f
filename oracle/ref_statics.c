/* oracle/ref_statics.c -- TEST INFRASTRUCTURE, not product code.
 *
 * Entry points onto `static` functions of the reference's bwamem.c: the file is compiled HERE, from where it lies under
 * /root/reference/lib/aln (the #include below resolves through -I$(REF); nothing of it is copied), so that its file-local
 * functions can be forwarded to like the exported ones of ref_shim.c.  This translation unit replaces the separate
 * compilation of bwamem.c in oracle/Makefile: its exported functions (mem_opt_init, mem_approx_mapq_se, bseq_bsconvert) are
 * the same code either way.
 */
#include "bwamem.c"

#define API __attribute__((visibility("default")))

/* read_clipping (bwamem.c:286-303) over read_identify_adaptor (:258-274) and clip_read_by_quality (:276-284):
 * out = l_adaptor, clip5, clip3, l_seq after clipping, offset of the clipped sequence in the original */
API void ref_read_clipping(int l_seq, const uint8_t *seq, const char *qual, const uint8_t *adaptor, int l_adaptor,
                           int clip5, int clip3, int min_base_qual, int out[5])
{
	mem_opt_t *o = mem_opt_init();
	bseq1_t s;
	memset(&s, 0, sizeof(s));
	o->clip5 = clip5; o->clip3 = clip3; o->min_base_qual = min_base_qual;
	s.l_seq = l_seq; s.seq = (uint8_t*)seq; s.qual = (char*)qual;
	read_clipping(&s, (uint8_t*)adaptor, l_adaptor, o);
	out[0] = s.l_adaptor; out[1] = s.clip5; out[2] = s.clip3; out[3] = s.l_seq; out[4] = (int)(s.seq - s.seq0);
	free(o);
}

/* check_paired_read_names (bwamem.c:210-216) returns for names it accepts; it ends the process for the others (err_fatal), so
 * only accepted pairs are passed here */
API void ref_check_paired_read_names(const char *n1, const char *n2) { check_paired_read_names(n1, n2); }
