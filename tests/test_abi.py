"""The C-ABI: libmfk.so loads without a GPU and exports every function include/mfk.h declares; struct layouts agree
between the header (as gcc sees it) and the ctypes / numpy mirrors.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

from mustache_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mfk.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mfk_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for n in ("mfk_create", "mfk_destroy", "mfk_last_error", "mfk_findflanks", "mfk_findflanks_fetch", "mfk_pairflanks",
              "mfk_pairflanks_fetch", "mfk_ir_detect", "mfk_profile_enable", "mfk_profile_fetch"):
        assert n in names


def test_hip_library_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in declared_functions():
        assert hasattr(so, n), n


def test_oracle_exports_the_same_signatures(oracle):
    for n in declared_functions():
        # timing hooks / device memory / RCCL / the host-table loops of the sharded path (stated in numpy in mustache_amd/dist.py) / the device decode (checked against zlib and the host packer): HIP library only
        if n.startswith(("mfk_profile_", "mfk_comm_", "mfk_allgather_", "mfk_flanks_", "mfk_flank_runs_", "mfk_pairs_")) or n in ("mfk_recall_noalloc", "mfk_span_timing", "mfk_bgzf_inflate", "mfk_bgzf_crc32", "mfk_bgzf_block_table", "mfk_decode_bam", "mfk_decode_bam_fetch"):
            continue
        assert hasattr(oracle.so, n.replace("mfk_", "mfko_", 1)), n


def test_struct_layouts(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(mfk_reads),sizeof(mfk_params),sizeof(mfk_site),sizeof(mfk_flank),sizeof(mfk_pair),'
                   'offsetof(mfk_flank,seq_off),offsetof(mfk_pair,ir5_off),sizeof(mfk_clip_rec),offsetof(mfk_reads,n_clip));return 0;}\n' % HEADER)
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert sizes[0] == ctypes.sizeof(lib.MfkReads)
    assert sizes[1] == ctypes.sizeof(lib.MfkParams)
    assert sizes[2] == lib.SITE_DTYPE.itemsize
    assert sizes[3] == lib.FLANK_DTYPE.itemsize
    assert sizes[4] == lib.PAIR_DTYPE.itemsize
    assert sizes[5] == lib.FLANK_DTYPE.fields["seq_off"][1]
    assert sizes[6] == lib.PAIR_DTYPE.fields["ir5_off"][1]
    from mustache_amd import pack
    assert sizes[7] == pack.CLIP_REC.itemsize == 16
    assert sizes[8] == lib.MfkReads.n_clip.offset       # the optional trailing fields sit where the old struct ended


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.MfkError):
        lib.load(0)
