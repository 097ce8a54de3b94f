// index_build.h -- the device-side FM-index builder (k_index.hip), called by shim.hip
#pragma once
#include <hip/hip_runtime.h>
#include "devbuf.hpp"

// FM index of one converted text ([fwd ; revcomp(fwd)] of the 2-bit genome d_pac, C>T if parent else G>A), built in HBM:
//   *bwt_out : the .bwt file body (occurrence blocks interleaved, lib/aln/bwt.h:93-101), 64 bytes of slack after it
//   *sa_out  : SA'[j * dense_intv], j = 0 .. n / dense_intv (entry 0 = -1), what the seeding kernels walk to
//   meta     : primary, L2, seq_len, bwt_size, sa_intv (= file_intv), n_sa
//   h_bwt / h_sa (optional, host): the file-format arrays (bwt_size words; n_sa entries at file_intv)
int bsx_ix_build_fmi(hipStream_t st, int n_cu, const uint8_t *d_pac, int64_t l_pac, int parent, int dense_intv, int file_intv,
                     DevBuf *bwt_out, DevBuf *sa_out, bsx_fmi_t *meta, uint32_t *h_bwt, uint64_t *h_sa);
