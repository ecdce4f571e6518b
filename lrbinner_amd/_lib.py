"""ctypes binding of liblrb_hip.so (include/lrb_hip.h).

There is deliberately no CPU fallback: if the HIP library has not been built
(``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C lrbinner_amd/csrc``) importing the product path fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "liblrb_hip.so")

LRB_OK = 0
ERR_NAMES = {1: "LRB_ERR_ARG", 2: "LRB_ERR_HIP", 3: "LRB_ERR_NOMEM", 4: "LRB_ERR_NODEVICE",
             5: "LRB_ERR_IO", 6: "LRB_ERR_FORMAT"}
K15_ENTRIES = 4 ** 15
K15_HALF_ENTRIES = 4 ** 15 // 2  # canonical half of the table (lrb_k15_fold_half_dev)
HIST_BINS = 60

vp = C.c_void_p
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)

# name -> (restype, argtypes); mirrors include/lrb_hip.h one to one
PROTOTYPES = {
    "lrb_last_error": (C.c_char_p, []),
    "lrb_version": (C.c_int, []),
    "lrb_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lrb_ctx_create": (C.c_int, [C.c_int, vp, C.c_int, C.POINTER(vp)]),
    "lrb_ctx_destroy": (C.c_int, [vp]),
    "lrb_ctx_sync": (C.c_int, [vp]),
    "lrb_ctx_trim": (C.c_int, [vp, C.c_uint64]),
    "lrb_ctx_list_pool": (C.c_int, [vp, C.c_uint64]),
    "lrb_ctx_ws_info": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64)]),
    "lrb_ctx_partition_retries": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
    "lrb_ctx_stream": (C.c_int, [vp, C.POINTER(vp)]),
    "lrb_dev_alloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "lrb_dev_free": (C.c_int, [vp, vp]),
    "lrb_dev_mem_info": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lrb_host_alloc": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "lrb_host_free": (C.c_int, [vp, vp]),
    "lrb_dev_memset": (C.c_int, [vp, vp, C.c_int, C.c_uint64]),
    "lrb_copy_h2d": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lrb_copy_d2h": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lrb_kmer_dim": (C.c_int, [C.c_int, u32p]),
    "lrb_kmer_lut": (C.c_int, [C.c_int, u32p, u32p]),
    "lrb_pack_layout": (C.c_int, [u64p, C.c_uint64, u32p, u64p, u64p]),
    "lrb_pack_reads_dev": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, vp, vp]),
    "lrb_kmer_counts_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, C.c_int, vp]),
    "lrb_planes_from_codes_dev": (C.c_int, [vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_kmer_counts3_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_int, vp]),
    "lrb_planes_t_layout": (C.c_int, [u32p, C.c_uint64, u32p, u64p]),
    "lrb_pack_planes_t_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_planes_t_from_planes_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_kmer_counts3t_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_codes_t_layout": (C.c_int, [u32p, C.c_uint64, u32p, u64p]),
    "lrb_codes_t_from_codes_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_kmer_counts4t_dev": (C.c_int, [vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_kmer_counts_t_dev": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_kmer_counts_host": (C.c_int, [vp, u8p, u64p, C.c_uint64, C.c_int, u32p]),
    "lrb_k15_accumulate_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_k15_accumulate_part_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint64, vp]),
    "lrb_k15_mirror_dev": (C.c_int, [vp, vp]),
    "lrb_k15_fold_half_dev": (C.c_int, [vp, vp, vp]),
    "lrb_k15_expand_half_dev": (C.c_int, [vp, vp, vp]),
    "lrb_rccl_unique_id": (C.c_int, [u8p]),
    "lrb_rccl_comm_create": (C.c_int, [vp, C.c_int, C.c_int, u8p, C.POINTER(vp)]),
    "lrb_rccl_comm_destroy": (C.c_int, [vp]),
    "lrb_k15_allreduce": (C.c_int, [vp, vp, vp, C.c_uint64]),
    "lrb_k15_accumulate_host": (C.c_int, [vp, u8p, u64p, C.c_uint64, vp]),
    "lrb_k15_write_file": (C.c_int, [vp, vp, C.c_char_p]),
    "lrb_k15_write_file_async": (C.c_int, [vp, vp, C.c_char_p, C.POINTER(vp)]),
    "lrb_k15_write_file_part_async": (C.c_int, [vp, vp, C.c_char_p, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "lrb_job_wait": (C.c_int, [vp]),
    "lrb_k15_read_file": (C.c_int, [vp, vp, C.c_char_p]),
    "lrb_cov_hist_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_int64, C.c_int,
                                   vp, vp]),
    "lrb_cov_map_build_dev": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "lrb_cov_hist_map_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_int, vp, vp]),
    "lrb_cov_hist_sweep_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp, C.c_int, vp, vp]),
    "lrb_k15_lists_geometry": (C.c_int, [vp, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "lrb_k15_lists_geometry_for": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "lrb_k15_lists_bounds_words": (C.c_uint64, [C.c_uint64]),
    "lrb_k15_lists_part_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp]),
    "lrb_k15_lists_tally_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp]),
    "lrb_k15_accumulate_half_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, vp]),
    "lrb_cov_map_build_half_dev": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "lrb_cov_lists_sweep_dev": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp, vp, vp, C.c_int, vp, vp]),
    "lrb_packed_cov_hist_many": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, vp, C.c_int]),
    "lrb_packed_lists_create": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, C.c_int, C.c_int, C.POINTER(vp)]),
    "lrb_winlists_valid": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
    "lrb_winlists_info": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "lrb_winlists_tally": (C.c_int, [vp, vp, vp]),
    "lrb_winlists_cov_hist": (C.c_int, [vp, vp, vp, C.c_int]),
    "lrb_winlists_free": (C.c_int, [vp, vp]),
    "lrb_packed_k15_accumulate_half": (C.c_int, [vp, vp, vp]),
    "lrb_cov_rows_text": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_int, vp, u32p]),
    "lrb_cov_hist_host": (C.c_int, [vp, u8p, u64p, C.c_uint64, vp, C.c_int64, C.c_int, u32p,
                                    u32p]),
    "lrb_packed_create": (C.c_int, [vp, u8p, u64p, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "lrb_packed_create_dev": (C.c_int, [vp, vp, u64p, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "lrb_packed_kmer_counts_dev": (C.c_int, [vp, vp, C.c_int, vp]),
    "lrb_packed_kmer_counts_many_dev": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, C.c_int, vp]),
    "lrb_packed_free": (C.c_int, [vp, vp]),
    "lrb_packed_info": (C.c_int, [vp, u64p, u64p]),
    "lrb_packed_kmer_counts": (C.c_int, [vp, vp, C.c_int, u32p]),
    "lrb_packed_k15_accumulate": (C.c_int, [vp, vp, vp]),
    "lrb_packed_k15_accumulate_many": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, vp]),
    "lrb_packed_k15_tally_half_many": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, vp]),
    "lrb_packed_k15_tally_half_many_for": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, vp, C.c_int]),
    "lrb_packed_lists_resident": (C.c_int, [vp, C.POINTER(vp), C.c_uint64, C.c_int, C.POINTER(C.c_int)]),
    "lrb_packed_group_starts": (C.c_int, [C.POINTER(vp), C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "lrb_packed_cov_hist": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, u32p, u32p]),
    "lrb_packed_kmer_text": (C.c_int, [vp, vp, C.c_int, vp, u32p]),
    "lrb_kmer_text_host": (C.c_int, [vp, u8p, u64p, C.c_uint64, C.c_int, u8p, u32p]),
    "lrb_cov_text_host": (C.c_int, [vp, u8p, u64p, C.c_uint64, vp, C.c_int64, C.c_int, u8p, u32p]),
    "lrb_packed_cov_text": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int, vp, u32p]),
    "lrb_seed_dist_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_uint64, vp]),
    "lrb_seed_hist_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint32, vp]),
    "lrb_gauss_assign_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, vp, C.c_int, vp, vp]),
    "lrb_hdb_core_dist_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, C.c_uint32, vp]),
    "lrb_hdb_mst_dev": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, u32p, u32p, C.POINTER(C.c_float),
                                  u32p]),
    "lrb_hdb_labels": (C.c_int, [C.c_uint64, u32p, u32p, C.POINTER(C.c_float), C.c_uint32,
                                 C.POINTER(C.c_int32), u32p]),
    "lrb_hdbscan_host": (C.c_int, [vp, C.POINTER(C.c_float), C.c_uint64, C.c_int, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_int32), u32p]),
    "lrb_hdbscan_host_ex": (C.c_int, [vp, C.POINTER(C.c_float), C.c_uint64, C.c_int, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(C.c_int32), u32p]),
    "lrb_mt_shuffle_i64": (C.c_int, [u32p, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.c_uint64]),
    "lrb_vae_create": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_float), C.c_float, C.c_float, C.c_uint64, C.POINTER(vp)]),
    "lrb_vae_destroy": (C.c_int, [vp]),
    "lrb_vae_sizes": (C.c_int, [vp, u64p, u64p]),
    "lrb_vae_set": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.c_uint64]),
    "lrb_vae_get": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.c_uint64]),
    "lrb_vae_steps_done": (C.c_int, [vp, u64p]),
    "lrb_vae_train_dev": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_int]),
    "lrb_vae_encode_dev": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "lrb_vae_debug_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float), C.c_uint64]),
    "lrb_reader_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "lrb_fasta_scan": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "lrb_fasta_records_view": (C.c_int, [vp, u64p, C.POINTER(u8p), C.POINTER(u64p), C.POINTER(u8p), C.POINTER(u64p)]),
    "lrb_fasta_write_fragments": (C.c_int, [vp, C.c_char_p, u64p, u32p]),
    "lrb_fasta_records_free": (C.c_int, [vp]),
    "lrb_reader_next": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.POINTER(u8p), C.POINTER(u64p),
                                  u64p]),
    "lrb_reader_close": (C.c_int, [vp]),
    "lrb_preader_open": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "lrb_preader_open_shard": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32,
                                         C.POINTER(vp)]),
    "lrb_preader_next": (C.c_int, [vp, C.POINTER(u8p), C.POINTER(u64p), u64p]),
    "lrb_preader_open_ex": (C.c_int, [C.c_char_p, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "lrb_preader_packed_view": (C.c_int, [vp, C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u64p), C.POINTER(u64p), C.POINTER(u32p)]),
    "lrb_pack_host_sizes": (C.c_int, [u64p, C.c_uint64, u64p, u64p]),
    "lrb_pack_reads_host": (C.c_int, [u8p, u64p, C.c_uint64, u32p, u32p, u64p, u64p, u32p]),
    "lrb_pack_reads_host_scalar": (C.c_int, [u8p, u64p, C.c_uint64, u32p, u32p, u64p, u64p, u32p]),
    "lrb_packed_create_packed": (C.c_int, [vp, u32p, u32p, u64p, u64p, u32p, u64p, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "lrb_preader_info": (C.c_int, [vp, C.POINTER(C.c_int), u64p, u64p]),
    "lrb_preader_close": (C.c_int, [vp]),
    "lrb_profile_text_bound": (C.c_uint64, [C.c_uint64, C.c_uint32]),
    "lrb_debug_format_f": (C.c_int, [C.c_double, C.c_char_p, C.c_char_p]),
    "lrb_format_com": (C.c_int, [u32p, u32p, C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_char_p,
                                 u64p, f64p]),
    "lrb_format_cov": (C.c_int, [u32p, u32p, C.c_uint64, C.c_uint32, C.c_int, C.c_char_p, u64p,
                                 f64p]),
    "lrb_com_row_bytes": (C.c_uint64, [C.c_uint32]),
    "lrb_cov_row_bytes": (C.c_uint64, [C.c_uint32]),
    "lrb_format_com_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp]),
    "lrb_format_cov_dev": (C.c_int, [vp, vp, vp, C.c_uint64, C.c_uint32, vp, vp]),
}


class LrbError(RuntimeError):
    """A C-ABI call returned a non-zero code."""

    def __init__(self, fn, code, msg):
        self.fn, self.code, self.msg = fn, code, msg
        super().__init__(f"{fn} -> {ERR_NAMES.get(code, code)}: {msg}")


_lib = None


def lib():
    """The loaded library.  Raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build() or `make -C lrbinner_amd/csrc`). "
                "lrbinner_amd has no CPU fallback.")
        # One HIP runtime per process: torch bundles its own libamdhip64 (same SONAME
        # as /opt/rocm's).  Importing torch first makes our NEEDED entry resolve to the
        # copy torch already loaded; the other order would load two runtimes and the
        # second one sees no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(fn_name, code):
    if code != LRB_OK:
        raise LrbError(fn_name, code, (lib().lrb_last_error() or b"").decode(errors="replace"))


def call(fn_name, *args):
    """Call an int-returning entry point and raise LrbError on failure."""
    check(fn_name, getattr(lib(), fn_name)(*args))
