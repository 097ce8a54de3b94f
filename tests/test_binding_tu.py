"""Boundary evidence in the build container: INTEGRATION.md section B's translation unit -- the binding a BISCUIT maintainer would add
behind mem_process_seqs (lib/aln/bwamem.h:184) -- is taken out of the document as it stands, compiled against the REFERENCE'S OWN headers
(-I/root/reference/lib/aln, no stand-ins) with -Wall -Werror, and layout assertions are compiled with it:

  mem_opt_t    <-> bsx_opt_t      (lib/aln/bwamem.h:54-124): size, and offset + size of every field
  mem_pestat_t <-> bsx_pestat_t   (lib/aln/bwamem.h:126-131): the cast `(const bsx_pestat_t*)pes0` of the binding
  bwtintv_t    <-> bsx_intv_t     (lib/aln/bwt.h:80-82)
  kswr_t       <-> bsx_sw_res_t   (lib/aln/ksw.h:14-19)
  bseq1_t      <-> bsx_read_t     (lib/aln/bwa.h:52-61): every field the binding copies has the same type width on both sides

Skipped where /root/reference is absent (the GPU box)."""
import os
import re
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/lib/aln"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers absent (not the build container)")

OPT_FIELDS = ("a b o_del e_del o_ins e_ins pen_unpaired pen_clip5 pen_clip3 w zdrop max_mem_intv T flag min_seed_len min_chain_weight "
              "max_chain_extend split_factor split_width max_occ max_chain_gap n_threads chunk_size mask_level drop_ratio XA_drop_ratio "
              "mask_level_redun mapQ_coef_len mapQ_coef_fac max_ins max_matesw max_XA_hits max_XA_hits_alt mat parent bsstrand ctmat gamat "
              "adaptor1 l_adaptor1 adaptor2 l_adaptor2 clip5 clip3 min_base_qual has_bc").split()
PES_FIELDS = "low high set failed avg std".split()
READ_FIELDS = "l_seq id name comment barcode umi qual sam seq seq0 l_seq0 l_adaptor clip5 clip3".split()


def binding_tu():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```c\n(/\* lib/aln/bwamem_gpu\.c.*?)```", md, re.S)
    assert m, "INTEGRATION.md section B no longer holds the binding translation unit"
    return m.group(1)


def asserts():
    lines = ["#include <stddef.h>", '#include "bwt.h"', '#include "ksw.h"', '#include "bwa.h"',
             "#define SAME(ta, tb, f) _Static_assert(offsetof(ta, f) == offsetof(tb, f) && sizeof(((ta*)0)->f) == sizeof(((tb*)0)->f), #ta \".\" #f \" vs \" #tb)",
             "#define SAMESZ(ta, tb, f) _Static_assert(sizeof(((ta*)0)->f) == sizeof(((tb*)0)->f), #ta \".\" #f \" width vs \" #tb)",
             "_Static_assert(sizeof(mem_opt_t) == sizeof(bsx_opt_t), \"mem_opt_t vs bsx_opt_t size\");",
             "_Static_assert(sizeof(mem_pestat_t) == sizeof(bsx_pestat_t), \"mem_pestat_t vs bsx_pestat_t size\");",
             "_Static_assert(sizeof(bwtintv_t) == sizeof(bsx_intv_t), \"bwtintv_t vs bsx_intv_t size\");",
             "_Static_assert(sizeof(kswr_t) == sizeof(bsx_sw_res_t), \"kswr_t vs bsx_sw_res_t size\");"]
    lines += ["SAME(mem_opt_t, bsx_opt_t, %s);" % f for f in OPT_FIELDS]
    lines += ["SAME(mem_pestat_t, bsx_pestat_t, %s);" % f for f in PES_FIELDS]
    lines += ["SAME(bwtintv_t, bsx_intv_t, x);", "SAME(bwtintv_t, bsx_intv_t, info);"]
    lines += ["SAME(kswr_t, bsx_sw_res_t, %s);" % f for f in "score te qe score2 te2 tb qb".split()]
    lines += ["SAMESZ(bseq1_t, bsx_read_t, %s);" % f for f in READ_FIELDS]   # (bseq1_t carries bisseq[2] between seq and seq0: offsets differ, the binding copies field by field)
    # the flag bits and kernel constants the two sides must agree on
    lines += ["_Static_assert(MEM_F_PE == BSX_F_PE && MEM_F_NO_MULTI == BSX_F_NO_MULTI && MEM_F_NO_RESCUE == BSX_F_NO_RESCUE && MEM_F_SELF_OVLP == BSX_F_SELF_OVLP, \"MEM_F_* vs BSX_F_*\");",
              "_Static_assert(KSW_XBYTE == BSX_KSW_XBYTE && KSW_XSTART == BSX_KSW_XSTART && KSW_XSUBO == BSX_KSW_XSUBO, \"KSW_X* vs BSX_KSW_X*\");"]
    return "\n".join(lines) + "\n"


def test_binding_translation_unit_compiles_against_the_reference_headers(tmp_path):
    src = tmp_path / "bwamem_gpu.c"
    src.write_text(binding_tu() + "\n/* ---- layout assertions (tests/test_binding_tu.py) ---- */\n" + asserts())
    cmd = ["gcc", "-std=gnu99", "-Wall", "-Werror", "-c", "-I" + REF, "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "bwamem_gpu.o")]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    # the object defines the reference's seam and binds only to symbols the product library exports
    nm = subprocess.run(["nm", str(tmp_path / "bwamem_gpu.o")], stdout=subprocess.PIPE).stdout.decode()
    assert re.search(r"\bT mem_process_seqs\b", nm) and re.search(r"\bT mem_gpu_init\b", nm)
    und = set(re.findall(r"\bU (bsx_\w+)", nm))
    assert und == {"bsx_index_load", "bsx_device_open", "bsx_device_upload_index", "bsx_opt_init", "bsx_process_seqs"}, und
    lib = os.path.join(ROOT, "biscuit_amd", "libbiscuit_amd.so")
    if os.path.exists(lib):
        exp = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE).stdout.decode()
        for s in und:
            assert re.search(r"\bT %s\b" % s, exp), s


def test_a_wrong_layout_is_caught(tmp_path):
    """the assertions bite: a bsx_pestat_t with its doubles first must not compile"""
    src = tmp_path / "bad.c"
    src.write_text('#include <stddef.h>\n#include "bwamem.h"\ntypedef struct { double avg, std; int low, high, set, failed; } bsx_pestat_t;\n'
                   "#define SAME(ta, tb, f) _Static_assert(offsetof(ta, f) == offsetof(tb, f) && sizeof(((ta*)0)->f) == sizeof(((tb*)0)->f), #f)\n"
                   "SAME(mem_pestat_t, bsx_pestat_t, low);\n")
    p = subprocess.run(["gcc", "-std=gnu99", "-c", "-I" + REF, str(src), "-o", str(tmp_path / "bad.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode != 0
