"""ctypes binding of libmi_stark.so (the C ABI declared in include/mi_stark.h).

Plumbing only: device memory comes from torch (int64 tensors used as raw u64 containers), the stream is
torch's current stream, and every call goes straight to the HIP library.  There is no fallback of any
kind: if the shared library is missing or no GPU is usable, loading / context creation raises.
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI_STARK_LIB") or os.path.join(HERE, "libmi_stark.so")   # override: A/B builds of the same library
P = 0xFFFFFFFF00000001

u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(u64)


class MiStarkError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmi_stark.so (no GPU needed for loading; compute entry points need one)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiStarkError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                "There is no CPU fallback.")
        # torch's HIP runtime first, when this process is going to use torch at all: the library binds to the libamdhip64 / libhsa-runtime64
        # that are already loaded, and a process that loaded the system's copy first and torch's bundled one later ends up with two HSA
        # runtimes -- hipGetDeviceCount then fails in whichever came second (seen on the GPU box: "no HIP device available")
        if "torch" not in sys.modules and os.environ.get("MI_STARK_NO_TORCH_PRELOAD", "0") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = ctypes.CDLL(LIB_PATH)
        L.mi_last_error.restype = ctypes.c_char_p
        L.mi_version.restype = ctypes.c_char_p
        L.mi_dev_alloc.restype = ctypes.c_void_p
        L.mi_dbg_host_mul.restype = u64
        L.mi_dbg_host_mul.argtypes = [u64, u64]
        _lib = L
    return _lib


# every symbol include/mi_stark.h declares (checked by tests/test_capi_symbols.py)
EXPORTS = [
    "mi_ctx_create", "mi_ctx_destroy", "mi_ctx_set_stream", "mi_ctx_sync", "mi_ctx_set_workspace_limit",
    "mi_last_error", "mi_version", "mi_device_count",
    "mi_ntt_dev", "mi_lde_dev", "mi_ntt", "mi_lde",
    "mi_poseidon_hash_full_result", "mi_poseidon_hash", "mi_transcript_put", "mi_poseidon_linear_hash", "mi_poseidon_permute_dev",
    "mi_linear_hash_rows_dev", "mi_linear_hash_absorb_dev", "mi_merkle_build_dev", "mi_merkle_levels_dev", "mi_merkle_build",
    "mi_merkle_group_proofs_dev", "mi_merkle_group_proofs_tiled_dev", "mi_lde_merkle_host_tiled", "mi_lde_merkle_dev_tiled", "mi_evmap_tiled_dev", "mi_untile_dev",
    "mi_fri_fold_dev", "mi_fri_fold_range_dev", "mi_fri_transpose_dev", "mi_q_split_dev", "mi_evmap_dev", "mi_evmap_range_dev", "mi_batch_inverse3_dev", "mi_calculate_h1h2_dev", "mi_calculate_z_dev", "mi_calculate_z_batch_dev",
    "mi_geom_seq_dev", "mi_geom_seq3_dev", "mi_x_div_x_sub_dev", "mi_zhinv",
    "mi_fill_synthetic_dev", "mi_fill_synthetic_2d_dev", "mi_copy_2d_dev", "mi_dev_alloc", "mi_dev_free", "mi_copy_h2d", "mi_copy_d2h", "mi_copy_h2d_2d", "mi_dev_zero",
    "mi_set_poseidon_variant", "mi_set_poseidon_coop_max", "mi_set_ntt_tile", "mi_set_lde_fuse", "mi_set_leaf_mode", "mi_timer_start", "mi_timer_stop", "mi_timer_elapsed_ms",
    "mi_dbg_field_ops_dev", "mi_dbg_host_poseidon_permute", "mi_dbg_host_mul", "mi_dbg_host_e3_mul", "mi_dbg_host_e3_inv",
    "mi_dbg_host_dft16", "mi_dbg_lincomb_cols_dev", "mi_dbg_ntt_colmajor_dev",
    "mi_ctx_lend_workspace", "mi_dev_mem_info", "mi_lde_merkle_host_keep", "mi_lde_merkle_host_keep_tiled", "mi_tile_major_dev", "mi_chelpers_set_tiled_section", "mi_chelpers_set_tiled_consts", "mi_get_host_pack_threads",
    "mi_ctx_device", "mi_multi_lend", "mi_multi_plan_debug", "mi_multi_create", "mi_multi_create2", "mi_multi_lead", "mi_multi_set_transient", "mi_multi_transient_need", "mi_multi_windowed_need", "mi_multi_check_stats", "mi_multi_own", "mi_vmm_reserve", "mi_vmm_back", "mi_vmm_allow_peer", "mi_vmm_backed_bytes", "mi_vmm_free", "mi_multi_destroy", "mi_multi_shards", "mi_multi_peer_access", "mi_multi_ctx", "mi_multi_set_pack_threads", "mi_multi_set_upload_mode", "mi_multi_set_row_images", "mi_multi_set_device", "mi_multi_copy", "mi_multi_sync", "mi_multi_last_upload_mode", "mi_multi_commit", "mi_multi_group_proofs",
    "mi_multi_tree_release_rows", "mi_multi_tree_free", "mi_multi_tree_info", "mi_multi_tree_nodes", "mi_multi_gather_rows", "mi_multi_last_stats",
    "mi_lde_merkle_host", "mi_set_host_pack_threads", "mi_host_register", "mi_host_unregister", "mi_set_chelpers_min_words", "mi_chelpers_compile", "mi_chelpers_compile_micro", "mi_chelpers_free", "mi_chelpers_stats", "mi_chelpers_run_dev", "mi_dbg_host_chelpers_run", "mi_chelpers_build_native", "mi_chelpers_precompile_shard", "mi_chelpers_lower_stats", "mi_dbg_host_chelpers_run_lowered", "mi_chelpers_native_stats", "mi_set_chelpers_batch_rows", "mi_chelpers_reserve",
]


def _check(status):
    if status != 0:
        raise MiStarkError(f"mi_stark status {status}: {lib().mi_last_error().decode()}")


def _hp(a):
    """host numpy uint64 array -> pointer"""
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_p64)


def _dp(t, offset_elems=0):
    """torch device tensor (int64 container) -> u64 pointer"""
    return ctypes.cast(ctypes.c_void_p(t.data_ptr() + 8 * offset_elems), _p64)


class Context:
    """One context per process / GPU.  Work is enqueued on torch's current stream."""

    def __init__(self, device=0, workspace_limit=None):
        import torch
        if not torch.cuda.is_available():
            raise MiStarkError("no GPU visible: mi_stark has no CPU fallback")
        self.torch = torch
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.h = ctypes.c_void_p()
        _check(lib().mi_ctx_create(ctypes.byref(self.h), ctypes.c_int(device)))
        self.use_torch_stream()
        if workspace_limit:
            _check(lib().mi_ctx_set_workspace_limit(self.h, u64(workspace_limit)))

    def use_torch_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().mi_ctx_set_stream(self.h, ctypes.c_void_p(s)))

    def close(self):
        if self.h:
            lib().mi_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def sync(self):
        _check(lib().mi_ctx_sync(self.h))

    # ---- memory helpers
    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.int64, device=self.device)

    def zeros(self, *shape):
        return self.torch.zeros(*shape, dtype=self.torch.int64, device=self.device)

    def to_device(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        return self.torch.from_numpy(a.view(np.int64)).to(self.device)

    def to_host(self, t):
        return t.detach().cpu().contiguous().numpy().view(np.uint64)

    def set_poseidon_variant(self, v):
        _check(lib().mi_set_poseidon_variant(self.h, ctypes.c_int(v)))

    def set_leaf_mode(self, line_aligned):
        _check(lib().mi_set_leaf_mode(self.h, ctypes.c_int(int(line_aligned))))

    def set_lde_fuse(self, fuse):
        _check(lib().mi_set_lde_fuse(self.h, ctypes.c_int(int(fuse))))

    def set_host_pack_threads(self, threads):
        """mi_lde_merkle_host: host threads that pack column chunks into page-locked staging (0 = strided 2-D copies)."""
        _check(lib().mi_set_host_pack_threads(self.h, ctypes.c_int(threads)))

    def set_chelpers_batch_rows(self, rows):
        _check(lib().mi_set_chelpers_batch_rows(self.h, u64(rows)))

    def set_chelpers_min_words(self, words):
        _check(lib().mi_set_chelpers_min_words(self.h, u64(words)))

    def transcript_put(self, tr, vals):
        """Transcript::put on a dict {state[4], pending[8], out[12], pending_cursor, out_cursor} (host numpy), one launch"""
        v = np.ascontiguousarray(vals, dtype=np.uint64).reshape(-1)
        pc, oc = ctypes.c_uint32(tr["pending_cursor"]), ctypes.c_uint32(tr["out_cursor"])
        _check(lib().mi_transcript_put(self.h, _hp(tr["state"]), _hp(tr["pending"]), _hp(tr["out"]), ctypes.byref(pc), ctypes.byref(oc),
                                       _hp(v) if v.size else None, u64(v.size)))
        tr["pending_cursor"], tr["out_cursor"] = pc.value, oc.value

    def set_poseidon_coop_max(self, max_states):
        """launches of at most this many independent permutations use the wave-cooperative kernel (0: never)"""
        _check(lib().mi_set_poseidon_coop_max(self.h, u64(max_states)))

    def set_ntt_tile(self, log_b):
        _check(lib().mi_set_ntt_tile(self.h, ctypes.c_int(log_b)))

    # ---- NTT / LDE (device resident)
    def ntt(self, dst, src, n, ncols, inverse=False, dst_pitch=None, src_pitch=None, dst_off=0, src_off=0):
        _check(lib().mi_ntt_dev(self.h, _dp(dst, dst_off), u64(dst_pitch or ncols), _dp(src, src_off),
                                u64(src_pitch or ncols), u64(n), u64(ncols), ctypes.c_int(int(inverse))))

    def lde(self, out, inp, n_ext, n, ncols, out_pitch=None, in_pitch=None, out_off=0, in_off=0):
        _check(lib().mi_lde_dev(self.h, _dp(out, out_off), u64(out_pitch or ncols), _dp(inp, in_off),
                                u64(in_pitch or ncols), u64(n_ext), u64(n), u64(ncols)))

    # ---- Poseidon / Merkle
    def permute(self, out, inp, count):
        _check(lib().mi_poseidon_permute_dev(self.h, _dp(out), _dp(inp), u64(count)))

    def linear_hash_rows(self, digests, src, ncols, nrows, pitch=None, src_off=0):
        _check(lib().mi_linear_hash_rows_dev(self.h, _dp(digests), _dp(src, src_off), u64(pitch or ncols), u64(ncols), u64(nrows)))

    def linear_hash_absorb(self, digests, windows, nrows, first, final):
        """windows: [(tensor, element_offset, width, pitch)] column windows absorbed in order (see mi_stark.h)."""
        k = len(windows)
        bases = (ctypes.c_void_p * k)(*[t.data_ptr() + 8 * off for (t, off, w, p) in windows])
        pitches = (ctypes.c_uint64 * k)(*[p for (t, off, w, p) in windows])
        widths = (ctypes.c_uint64 * k)(*[w for (t, off, w, p) in windows])
        _check(lib().mi_linear_hash_absorb_dev(self.h, _dp(digests), ctypes.c_uint32(k), bases, pitches, widths, u64(nrows),
                                               ctypes.c_int(int(first)), ctypes.c_int(int(final))))

    def merkle_build(self, nodes, src, ncols, nrows, pitch=None, src_off=0):
        _check(lib().mi_merkle_build_dev(self.h, _dp(nodes), _dp(src, src_off), u64(pitch or ncols), u64(ncols), u64(nrows)))

    def merkle_levels(self, nodes, nleaves):
        _check(lib().mi_merkle_levels_dev(self.h, _dp(nodes), u64(nleaves)))

    def merkle_group_proofs(self, proofs, nodes, src, height, width, idx, pitch=None):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        _check(lib().mi_merkle_group_proofs_dev(self.h, _dp(proofs), _dp(nodes), _dp(src), u64(pitch or width),
                                                u64(height), u64(width), _hp(idx), u64(idx.size)))

    def lde_merkle_host(self, nodes, ext, host_trace_ptr, n, n_ext, ncols, ext_pitch=None, chunk_cols=0):
        """host_trace_ptr: address of the row-major n x ncols host trace (e.g. a pinned torch tensor's data_ptr())."""
        _check(lib().mi_lde_merkle_host(self.h, _dp(nodes), _dp(ext), u64(ext_pitch or ncols), ctypes.c_void_p(host_trace_ptr),
                                        u64(n), u64(n_ext), u64(ncols), u64(chunk_cols)))

    def lde_merkle_host_keep(self, nodes, ext, base, host_trace_ptr, n, n_ext, ncols, ext_pitch=None, base_pitch=None, chunk_cols=0):
        """lde_merkle_host, and the uploaded base-domain section stays in `base` (device, n x ncols) as well."""
        _check(lib().mi_lde_merkle_host_keep(self.h, _dp(nodes), _dp(ext), u64(ext_pitch or ncols), _dp(base), u64(base_pitch or ncols),
                                             ctypes.c_void_p(host_trace_ptr), u64(n), u64(n_ext), u64(ncols), u64(chunk_cols)))

    def lde_merkle_host_keep_tiled(self, nodes, ext, base_tiled, host_trace_ptr, n, n_ext, ncols, ext_pitch=None, chunk_cols=0):
        """lde_merkle_host_keep with the base-domain section kept TILE-MAJOR ([n / 64][ncols][64]: ChelpersProgram.set_tiled_section)."""
        _check(lib().mi_lde_merkle_host_keep_tiled(self.h, _dp(nodes), _dp(ext), u64(ext_pitch or ncols), _dp(base_tiled),
                                                   ctypes.c_void_p(host_trace_ptr), u64(n), u64(n_ext), u64(ncols), u64(chunk_cols)))

    def lde_merkle_host_tiled(self, nodes, ext_tiled, base, host_trace_ptr, n, n_ext, ncols, base_pitch=0, chunk_cols=0):
        """lde_merkle_host with the extension left TILE-MAJOR ([rows / 64][ncols][64]); base (or None): the trace itself, tile-major too
        (base_pitch 0) or row-major at base_pitch."""
        _check(lib().mi_lde_merkle_host_tiled(self.h, _dp(nodes), _dp(ext_tiled), _dp(base) if base is not None else None, u64(base_pitch),
                                              ctypes.c_void_p(host_trace_ptr), u64(n), u64(n_ext), u64(ncols), u64(chunk_cols)))

    def lde_merkle_dev_tiled(self, nodes, ext_tiled, src, n, n_ext, ncols, src_pitch=None, src_off=0):
        """extendPol + merkelize of a device-resident row-major section, the extension left TILE-MAJOR."""
        _check(lib().mi_lde_merkle_dev_tiled(self.h, _dp(nodes), _dp(ext_tiled), _dp(src, src_off), u64(src_pitch or ncols), u64(n), u64(n_ext), u64(ncols)))

    def merkle_group_proofs_tiled(self, proofs, nodes, src_tiled, ncols_total, height, width, idx):
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        _check(lib().mi_merkle_group_proofs_tiled_dev(self.h, _dp(proofs), _dp(nodes), _dp(src_tiled), u64(ncols_total), u64(height), u64(width),
                                                      _hp(idx), u64(idx.size)))

    def untile(self, dst, src_tiled, ncols_total, nrows_total, col0=0, row0=0, nrows=None, ncols=None, dst_pitch=None):
        """dst (row-major) <- rows [row0, row0 + nrows) x columns [col0, col0 + ncols) of a tile-major section."""
        nrows = nrows_total - row0 if nrows is None else nrows
        ncols = ncols_total - col0 if ncols is None else ncols
        _check(lib().mi_untile_dev(self.h, _dp(dst), u64(dst_pitch or ncols), _dp(src_tiled), u64(ncols_total), u64(nrows_total), u64(col0), u64(row0),
                                   u64(nrows), u64(ncols)))

    def tile_major(self, dst, ncols_total, col0, src, nrows, ncols, src_pitch=None):
        """dst (tile-major, nrows x ncols_total) <- src (row-major device tensor) at column col0 of the tiles, canonical."""
        _check(lib().mi_tile_major_dev(self.h, _dp(dst), u64(ncols_total), u64(col0), _dp(src), u64(src_pitch or ncols), u64(nrows), u64(ncols)))

    def lend_workspace(self, buf):
        """NTT / LDE scratch out of a caller-owned device tensor (None: back to the context's own workspace)."""
        if buf is None:
            _check(lib().mi_ctx_lend_workspace(self.h, ctypes.c_void_p(0), u64(0)))
        else:
            _check(lib().mi_ctx_lend_workspace(self.h, _dp(buf), u64(buf.numel() * 8)))

    def mem_info(self):
        f, t = u64(0), u64(0)
        _check(lib().mi_dev_mem_info(self.h, ctypes.byref(f), ctypes.byref(t)))
        return f.value, t.value

    def host_pack_threads(self):
        return int(lib().mi_get_host_pack_threads(self.h))

    def host_register(self, ptr, nbytes):
        _check(lib().mi_host_register(self.h, ctypes.c_void_p(ptr), u64(nbytes)))

    def host_unregister(self, ptr):
        _check(lib().mi_host_unregister(self.h, ctypes.c_void_p(ptr)))

    # ---- sparse device memory (mi_vmm_*): an address range, physical memory under the parts that are used
    def vmm_reserve(self, nbytes):
        base = ctypes.c_void_p()
        _check(lib().mi_vmm_reserve(self.h, u64(nbytes), ctypes.byref(base)))
        return int(base.value)

    def vmm_back(self, base, offset, nbytes):
        _check(lib().mi_vmm_back(self.h, ctypes.c_void_p(base), u64(offset), u64(nbytes)))

    def vmm_backed_bytes(self, base):
        out = ctypes.c_uint64()
        _check(lib().mi_vmm_backed_bytes(self.h, ctypes.c_void_p(base), ctypes.byref(out)))
        return int(out.value)

    def vmm_free(self, base):
        _check(lib().mi_vmm_free(self.h, ctypes.c_void_p(base)))

    def copy_h2d(self, dst_ptr, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        _check(lib().mi_copy_h2d(self.h, ctypes.c_void_p(dst_ptr), _hp(host.reshape(-1)), u64(host.size * 8)))

    def copy_d2h(self, src_ptr, count):
        out = np.empty(count, dtype=np.uint64)
        _check(lib().mi_copy_d2h(self.h, _hp(out), ctypes.c_void_p(src_ptr), u64(count * 8)))
        return out

    # ---- host-pointer (drop-in) variants
    def ntt_host(self, src, n, ncols, inverse=False):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        dst = np.empty(n * ncols, dtype=np.uint64)
        _check(lib().mi_ntt(self.h, _hp(dst), _hp(src.reshape(-1)), u64(n), u64(ncols), ctypes.c_int(int(inverse))))
        return dst.reshape(n, ncols)

    def lde_host(self, src, n_ext, n, ncols):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        out = np.empty(n_ext * ncols, dtype=np.uint64)
        _check(lib().mi_lde(self.h, _hp(out), _hp(src.reshape(-1)), u64(n_ext), u64(n), u64(ncols)))
        return out.reshape(n_ext, ncols)

    def merkle_build_host(self, src, ncols, nrows):
        src = np.ascontiguousarray(src, dtype=np.uint64)
        nodes = np.empty((2 * nrows - 1) * 4, dtype=np.uint64)
        _check(lib().mi_merkle_build(self.h, _hp(nodes), _hp(src.reshape(-1)), u64(ncols), u64(nrows)))
        return nodes

    def hash_full_result(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.uint64)
        out = np.empty(12, dtype=np.uint64)
        _check(lib().mi_poseidon_hash_full_result(self.h, _hp(out), _hp(inp)))
        return out

    def hash(self, inp):
        inp = np.ascontiguousarray(inp, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        _check(lib().mi_poseidon_hash(self.h, _hp(out), _hp(inp)))
        return out

    def linear_hash(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.uint64)
        out = np.empty(4, dtype=np.uint64)
        _check(lib().mi_poseidon_linear_hash(self.h, _hp(out), _hp(vals) if vals.size else None, u64(vals.size)))
        return out

    # ---- FRI and the rest
    def fri_fold(self, out, pol, prev_bits, cur_bits, nbits_ext, x):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        _check(lib().mi_fri_fold_dev(self.h, _dp(out), _dp(pol), ctypes.c_uint(prev_bits), ctypes.c_uint(cur_bits),
                                     ctypes.c_uint(nbits_ext), _hp(x)))

    def fri_fold_range(self, out, pol, prev_bits, cur_bits, nbits_ext, x, g0, g_count):
        x = np.ascontiguousarray(x, dtype=np.uint64)
        _check(lib().mi_fri_fold_range_dev(self.h, _dp(out), _dp(pol), ctypes.c_uint(prev_bits), ctypes.c_uint(cur_bits),
                                           ctypes.c_uint(nbits_ext), _hp(x), u64(g0), u64(g_count)))

    def merkle_paths(self, paths, nodes, height, idx):
        """sibling paths only (levels x 4 per query) of the tree in `nodes`"""
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        _check(lib().mi_merkle_group_proofs_dev(self.h, _dp(paths), _dp(nodes), None, u64(0), u64(height), u64(0), _hp(idx), u64(idx.size)))

    def fri_transpose(self, aux, pol, degree, tbits):
        _check(lib().mi_fri_transpose_dev(self.h, _dp(aux), _dp(pol), u64(degree), ctypes.c_uint(tbits)))

    def q_split(self, qq2, qq1, n, n_ext, qdeg):
        _check(lib().mi_q_split_dev(self.h, _dp(qq2), _dp(qq1), u64(n), u64(n_ext), ctypes.c_uint(qdeg)))

    def evmap(self, evals, pols, prime, lev, lpev, n, ext_bits, row0=None, nrows=None, tile_cols=None):
        """pols: list of (tensor, offset_elems, dim, stride); row0 / nrows: only the partial sums over those rows of the base domain;
        tile_cols: per polynomial, the width of its TILE-MAJOR section (offset_elems = 64 * column then) or 0"""
        k = len(pols)
        if tile_cols is not None:
            tc = np.ascontiguousarray(tile_cols, dtype=np.uint64)
            ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() + 8 * off for (t, off, _, _) in pols])
            dims = np.array([d for (_, _, d, _) in pols], dtype=np.uint32)
            strides = np.array([s for (_, _, _, s) in pols], dtype=np.uint64)
            pr = np.ascontiguousarray(prime, dtype=np.uint8)
            _check(lib().mi_evmap_tiled_dev(self.h, _dp(evals), u64(k), u64(n), ctypes.c_uint(ext_bits), ptrs,
                                            dims.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _hp(strides),
                                            pr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _hp(tc), _dp(lev), _dp(lpev)))
            return
        ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() + 8 * off for (t, off, _, _) in pols])
        dims = np.array([d for (_, _, d, _) in pols], dtype=np.uint32)
        strides = np.array([s for (_, _, _, s) in pols], dtype=np.uint64)
        pr = np.ascontiguousarray(prime, dtype=np.uint8)
        if row0 is not None:
            _check(lib().mi_evmap_range_dev(self.h, _dp(evals), u64(k), u64(n), ctypes.c_uint(ext_bits), ptrs,
                                            dims.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _hp(strides),
                                            pr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _dp(lev), _dp(lpev), u64(row0), u64(nrows)))
            return
        _check(lib().mi_evmap_dev(self.h, _dp(evals), u64(k), u64(n), ctypes.c_uint(ext_bits), ptrs,
                                  dims.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), _hp(strides),
                                  pr.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), _dp(lev), _dp(lpev)))

    def batch_inverse3(self, res, src, n):
        _check(lib().mi_batch_inverse3_dev(self.h, _dp(res), _dp(src), u64(n)))

    def calculate_h1h2(self, h1, h1_stride, h2, h2_stride, f, f_stride, t, t_stride, dim, n):
        """Polinomial::calculateH1H2 (polinomial.hpp:303-584) on strided device views (tensors whose first element is row 0 of the
        polynomial); raises MiStarkError ("number not included: w=<row>") when a row of f is not in t."""
        _check(lib().mi_calculate_h1h2_dev(self.h, _dp(h1), u64(h1_stride), _dp(h2), u64(h2_stride), _dp(f), u64(f_stride), _dp(t),
                                           u64(t_stride), ctypes.c_uint(dim), u64(n)))

    def calculate_z(self, z, z_stride, num, num_stride, den, den_stride, n):
        """Polinomial::calculateZ (polinomial.hpp:586-607) on strided device views; returns whether the product closes (== 1)."""
        closes = ctypes.c_int(0)
        _check(lib().mi_calculate_z_dev(self.h, _dp(z), u64(z_stride), _dp(num), u64(num_stride), _dp(den), u64(den_stride), u64(n),
                                        ctypes.byref(closes)))
        return bool(closes.value)

    def calculate_z_batch(self, products, n):
        """The grand products of one stage in one pass (mi_calculate_z_batch_dev); products = [(z, z_stride, num, num_stride, den,
        den_stride), ...] of strided device views over the same n rows.  Returns the list of closes flags."""
        k = len(products)
        P, U = ctypes.c_void_p * k, ctypes.c_uint64 * k
        z, zs = P(*[p[0].data_ptr() for p in products]), U(*[p[1] for p in products])
        nu, ns = P(*[p[2].data_ptr() for p in products]), U(*[p[3] for p in products])
        de, ds = P(*[p[4].data_ptr() for p in products]), U(*[p[5] for p in products])
        closes = (ctypes.c_int * k)()
        _check(lib().mi_calculate_z_batch_dev(self.h, ctypes.c_uint(k), z, zs, nu, ns, de, ds, u64(n), closes))
        return [bool(c) for c in closes]

    def geom_seq(self, out, n, start, ratio):
        _check(lib().mi_geom_seq_dev(self.h, _dp(out), u64(n), u64(start), u64(ratio)))

    def geom_seq3(self, out, n, ratio):
        r = np.ascontiguousarray(ratio, dtype=np.uint64)
        _check(lib().mi_geom_seq3_dev(self.h, _dp(out), u64(n), _hp(r)))

    def x_div_x_sub(self, out, x, n, xi):
        e = np.ascontiguousarray(xi, dtype=np.uint64)
        _check(lib().mi_x_div_x_sub_dev(self.h, _dp(out), _dp(x), u64(n), _hp(e)))

    def zhinv(self, nbits, nbits_ext):
        out = np.empty(1 << (nbits_ext - nbits), dtype=np.uint64)
        _check(lib().mi_zhinv(self.h, _hp(out), ctypes.c_uint(nbits), ctypes.c_uint(nbits_ext)))
        return out

    def fill_synthetic(self, out, count, seed, off=0):
        _check(lib().mi_fill_synthetic_dev(self.h, _dp(out, off), u64(count), u64(seed)))

    def fill_synthetic_2d(self, out, nrows, ncols, global_cols, col0, seed, out_pitch=None, out_off=0):
        _check(lib().mi_fill_synthetic_2d_dev(self.h, _dp(out, out_off), u64(out_pitch or ncols), u64(nrows), u64(ncols),
                                              u64(global_cols), u64(col0), u64(seed)))

    def copy_2d(self, dst, src, nrows, ncols, dst_pitch, src_pitch, dst_off=0, src_off=0):
        _check(lib().mi_copy_2d_dev(self.h, _dp(dst, dst_off), u64(dst_pitch), _dp(src, src_off), u64(src_pitch),
                                    u64(nrows), u64(ncols)))

    def dbg_field_ops(self, out, a, b, n):
        _check(lib().mi_dbg_field_ops_dev(self.h, _dp(out), _dp(a), _dp(b), u64(n)))

    def dbg_lincomb_cols(self, out, src, nrows, ncols, coef, pitch=None, src_off=0, coef_off=0, accumulate=False):
        _check(lib().mi_dbg_lincomb_cols_dev(self.h, _dp(out), _dp(src, src_off), u64(pitch or ncols), u64(nrows), u64(ncols),
                                             _dp(coef, coef_off), ctypes.c_int(int(accumulate))))

    # ---- timers (HIP events on the context's stream)
    def timer_start(self, slot=0):
        _check(lib().mi_timer_start(self.h, ctypes.c_int(slot)))

    def timer_stop(self, slot=0):
        _check(lib().mi_timer_stop(self.h, ctypes.c_int(slot)))

    def timer_ms(self, slot=0):
        ms = ctypes.c_float()
        _check(lib().mi_timer_elapsed_ms(self.h, ctypes.c_int(slot), ctypes.byref(ms)))
        return ms.value


class ChelpersParams(ctypes.Structure):
    """mi_chelpers_params (include/mi_stark.h)"""
    _fields_ = [("pols", ctypes.c_void_p), ("const_pols", ctypes.c_void_p), ("n_const", u64),
                ("challenges", ctypes.c_void_p), ("n_challenges", u64), ("publics", ctypes.c_void_p), ("n_publics", u64),
                ("x", ctypes.c_void_p), ("x_stride", u64), ("zhinv", ctypes.c_void_p), ("n_zhinv", u64), ("q", ctypes.c_void_p),
                ("evals", ctypes.c_void_p), ("n_evals", u64), ("xdiv", ctypes.c_void_p), ("xdivw", ctypes.c_void_p), ("f", ctypes.c_void_p)]


MI_CHELPERS_STEP42NS = 42
MI_CHELPERS_STEP52NS = 52
MI_CHELPERS_STEP2PREV, MI_CHELPERS_STEP3PREV, MI_CHELPERS_STEP3 = 20, 30, 31   # the base-domain steps: results go into pols


def default_chelpers_cache():
    """Code objects of compiled constraint programs, next to the library (in-tree, so a cache filled on the build machine travels)."""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "_chelpers_cache")


class ChelpersProgram:
    """A constraint-evaluator program (the reference's generated op / args tables) translated for the GPU.
    ctx = None compiles for the host debug executor only (no GPU needed)."""

    STAT_NAMES = ("opcodes", "field_ops", "after_copy_forwarding", "instructions_per_row", "live_words_as_generated",
                  "live_words_rescheduled", "base_temps", "ext_temps", "device_instructions_per_row", "lds_base_temps", "lds_ext_temps",
                  "spilled_base_temps", "temp_reads_per_row", "spill_reads_per_row", "lds_bytes_per_workgroup", "staged_columns")

    def __init__(self, ctx, ops, args, sections=(), n_const=0, nrows_ext=0, step=MI_CHELPERS_STEP42NS):
        """sections: [(element offset in pols, columns, rows)] the program reads (needed for the GPU form only)."""
        self.ctx = ctx
        ops = np.ascontiguousarray(ops, dtype=np.uint64)
        args = np.ascontiguousarray(args, dtype=np.uint64)
        sec = np.ascontiguousarray(np.array(list(sections), dtype=np.uint64).reshape(-1, 3))
        self.h = ctypes.c_void_p()
        _check(lib().mi_chelpers_compile(ctx.h if ctx is not None else None, ctypes.byref(self.h), ctypes.c_int(step), _hp(ops), u64(ops.size),
                                         _hp(args) if args.size else None, u64(args.size), _hp(sec.reshape(-1)) if sec.size else None,
                                         u64(sec.shape[0]), u64(n_const), u64(nrows_ext)))
        st = np.zeros(16, dtype=np.uint64)
        _check(lib().mi_chelpers_stats(self.h, _hp(st)))
        self.stats = {k: int(v) for k, v in zip(self.STAT_NAMES, st)}

    # operand kinds / operation classes of mi_chelpers_compile_micro (mi_stark.h: MI_CHP_*)
    K = dict(NONE=0, T1=1, T3=2, POL=3, POLS=4, NUM=5, CONST=6, CONSTS=7, CHAL=8, PUB=9, POL3=10, POL3S=11, X=12, ZHINV=13, Q=14, EVAL=15, XD=16, XDW=17,
             DPOL=18, DPOLS=19)
    C = dict(ADD=0, SUB=1, MUL=2, COPY=3, STOREQ=4, STOREF=5, STOREP=6)

    @classmethod
    def from_microops(cls, ctx, microops, sections=(), n_const=0, nrows_ext=0, step=MI_CHELPERS_STEP42NS):
        """The program as field operations (mi_chelpers_compile_micro: what host/steps_tracer.hpp records from per-row Steps code):
        microops = [(class, dst kind, dst slot, (kind, [words]), (kind, [words]) or None)], kinds / classes by name (K / C) or number."""
        class Operand(ctypes.Structure):
            _fields_ = [("kind", ctypes.c_uint32), ("reserved", ctypes.c_uint32), ("v", ctypes.c_uint64 * 4)]

        class MicroOp(ctypes.Structure):
            _fields_ = [("cls", ctypes.c_uint32), ("dst_kind", ctypes.c_uint32), ("dst_slot", ctypes.c_uint64), ("a", Operand), ("b", Operand)]

        def num(table, v):
            return table[v] if isinstance(v, str) else int(v)
        arr = (MicroOp * max(len(microops), 1))()
        for m, (c_, dk, slot, a, b) in zip(arr, microops):
            m.cls, m.dst_kind, m.dst_slot = num(cls.C, c_), num(cls.K, dk), int(slot)
            for o, src in ((m.a, a), (m.b, b)):
                if src is None:
                    continue
                o.kind = num(cls.K, src[0])
                for j, w in enumerate(src[1]):
                    o.v[j] = int(w)
        self = cls.__new__(cls)
        self.ctx = ctx
        sec = np.ascontiguousarray(np.array(list(sections), dtype=np.uint64).reshape(-1, 3))
        self.h = ctypes.c_void_p()
        _check(lib().mi_chelpers_compile_micro(ctx.h if ctx is not None else None, ctypes.byref(self.h), ctypes.c_int(step), arr, u64(len(microops)),
                                               _hp(sec.reshape(-1)) if sec.size else None, u64(sec.shape[0]), u64(n_const), u64(nrows_ext)))
        st = np.zeros(16, dtype=np.uint64)
        _check(lib().mi_chelpers_stats(self.h, _hp(st)))
        self.stats = {k: int(v) for k, v in zip(self.STAT_NAMES, st)}
        return self

    def set_tiled_section(self, offset):
        """The section at this offset lies tile-major in HBM and is read in place (before build_native / precompile_shard)."""
        _check(lib().mi_chelpers_set_tiled_section(self.h, u64(offset)))

    NATIVE_STAT_NAMES = ("kernels", "code_bytes", "build_ms", "cache_hits", "estimated_valu_per_row", "spill_words_moved_per_row",
                         "horner_chain_steps", "constant_words")

    def set_tiled_consts(self):
        """The constant polynomials handed to run_* are TILE-MAJOR ([nrows / 64][n_const][64], rows bit-reversed inside a tile)."""
        _check(lib().mi_chelpers_set_tiled_consts(self.h))

    def reserve(self, nrows):
        _check(lib().mi_chelpers_reserve(self.ctx.h, self.h, u64(nrows)))

    def lower_stats(self, chunk_cost=0):
        st = np.zeros(12, dtype=np.uint64)
        _check(lib().mi_chelpers_lower_stats(self.h, u64(chunk_cost), _hp(st)))
        return dict(zip(("kernels", "horner_chain_steps", "chain_pieces", "estimated_valu_per_row", "chain_coefficients", "piece_constants",
                         "folded_leaves", "spill_words_moved_per_row", "operand_loads_per_row", "distinct_operands", "linear_terms", "linear_sums"),
                        (int(v) for v in st)))

    def precompile_shard(self, shard, nshards, cache_dir=None, chunk_cost=0):
        """One process's share of a parallel build: fills the cache, keeps nothing."""
        if cache_dir is None:
            cache_dir = os.environ.get("MI_CHELPERS_CACHE", default_chelpers_cache())
        _check(lib().mi_chelpers_precompile_shard(self.h, cache_dir.encode(), u64(chunk_cost), ctypes.c_uint32(shard), ctypes.c_uint32(nshards)))

    def build_native(self, cache_dir=None, chunk_cost=0):
        """Compile the translated program to gfx950 code (no GPU needed); run / run52 then use the compiled kernels.
        cache_dir None = the in-tree cache next to the library (or $MI_CHELPERS_CACHE)."""
        if cache_dir is None:
            cache_dir = os.environ.get("MI_CHELPERS_CACHE", default_chelpers_cache())
        _check(lib().mi_chelpers_build_native(self.h, cache_dir.encode() if cache_dir else None, u64(chunk_cost)))
        st = np.zeros(8, dtype=np.uint64)
        _check(lib().mi_chelpers_native_stats(self.h, _hp(st)))
        self.native_stats = {k: int(v) for k, v in zip(self.NATIVE_STAT_NAMES, st)}
        return self.native_stats

    def close(self):
        if self.h:
            lib().mi_chelpers_free(self.ctx.h if self.ctx is not None else None, self.h)
            self.h = ctypes.c_void_p()

    @staticmethod
    def _params(pols_ptr, cpols_ptr, n_const, challenges, publics, x_ptr, x_stride, zhinv, q_ptr, keep, evals=(), xdiv_ptr=None, xdivw_ptr=None, f_ptr=None):
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1)
        pb = np.ascontiguousarray(publics, dtype=np.uint64).reshape(-1)
        zh = np.ascontiguousarray(zhinv, dtype=np.uint64).reshape(-1)
        ev = np.ascontiguousarray(evals, dtype=np.uint64).reshape(-1)
        keep.extend([ch, pb, zh, ev])
        return ChelpersParams(pols_ptr, cpols_ptr, n_const, ch.ctypes.data if ch.size else None, ch.size // 3,
                              pb.ctypes.data if pb.size else None, pb.size, x_ptr, x_stride, zh.ctypes.data if zh.size else None, zh.size, q_ptr,
                              ev.ctypes.data if ev.size else None, ev.size // 3, xdiv_ptr, xdivw_ptr, f_ptr)

    def run52(self, pols, const_pols, n_const, challenges, evals, xdiv, xdivw, f, row0, nrows):
        """step52ns: pols / const_pols / xdiv / xdivw / f device tensors; challenges / evals host arrays."""
        keep = []
        P = self._params(pols.data_ptr(), const_pols.data_ptr() if const_pols is not None else None, n_const, challenges, (), None, 0, (), None, keep,
                         evals=evals, xdiv_ptr=xdiv.data_ptr(), xdivw_ptr=xdivw.data_ptr(), f_ptr=f.data_ptr())
        _check(lib().mi_chelpers_run_dev(self.ctx.h, self.h, ctypes.byref(P), u64(row0), u64(nrows)))

    def run52_host(self, pols, const_pols, n_const, challenges, evals, xdiv, xdivw, f, rows):
        keep = []
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        P = self._params(pols.ctypes.data, const_pols.ctypes.data, n_const, challenges, (), None, 0, (), None, keep,
                         evals=evals, xdiv_ptr=xdiv.ctypes.data, xdivw_ptr=xdivw.ctypes.data, f_ptr=f.ctypes.data)
        _check(lib().mi_dbg_host_chelpers_run(self.h, ctypes.byref(P), _hp(rows), u64(rows.size)))

    def run(self, pols, const_pols, n_const, challenges, publics, x, x_stride, zhinv, q, row0, nrows):
        """pols / const_pols / x / q: device tensors (int64 containers); challenges / publics / zhinv: host arrays."""
        keep = []
        P = self._params(pols.data_ptr(), const_pols.data_ptr(), n_const, challenges, publics, x.data_ptr(), x_stride, zhinv, q.data_ptr(), keep)
        _check(lib().mi_chelpers_run_dev(self.ctx.h, self.h, ctypes.byref(P), u64(row0), u64(nrows)))

    def run_lowered_host(self, pols, const_pols, n_const, challenges, publics, x, x_stride, zhinv, q, rows, chunk_cost=0):
        """step42ns, the program as lowered for the native backend, on the CPU (test hook)."""
        keep = []
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        P = self._params(pols.ctypes.data, const_pols.ctypes.data, n_const, challenges, publics, x.ctypes.data, x_stride, zhinv, q.ctypes.data, keep)
        _check(lib().mi_dbg_host_chelpers_run_lowered(self.h, ctypes.byref(P), _hp(rows), u64(rows.size), u64(chunk_cost)))

    def run52_lowered_host(self, pols, const_pols, n_const, challenges, evals, xdiv, xdivw, f, rows, chunk_cost=0):
        keep = []
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        P = self._params(pols.ctypes.data, const_pols.ctypes.data, n_const, challenges, (), None, 0, (), None, keep,
                         evals=evals, xdiv_ptr=xdiv.ctypes.data, xdivw_ptr=xdivw.ctypes.data, f_ptr=f.ctypes.data)
        _check(lib().mi_dbg_host_chelpers_run_lowered(self.h, ctypes.byref(P), _hp(rows), u64(rows.size), u64(chunk_cost)))

    def run_base(self, pols, const_pols, n_const, challenges, publics, x, x_stride, row0, nrows):
        """step2prev / step3prev / step3: pols (device) is read and written; needs build_native."""
        keep = []
        P = self._params(pols.data_ptr(), const_pols.data_ptr(), n_const, challenges, publics, x.data_ptr(), x_stride, (), None, keep)
        _check(lib().mi_chelpers_run_dev(self.ctx.h, self.h, ctypes.byref(P), u64(row0), u64(nrows)))

    def run_base_host(self, pols, const_pols, n_const, challenges, publics, x, x_stride, rows, lowered=False, chunk_cost=0):
        """The same on the CPU (test hook; host numpy arrays, pols is written): the translated program, or -- lowered -- the program
        as the native backend lowers it."""
        keep = []
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        P = self._params(pols.ctypes.data, const_pols.ctypes.data, n_const, challenges, publics, x.ctypes.data, x_stride, (), None, keep)
        if lowered:
            _check(lib().mi_dbg_host_chelpers_run_lowered(self.h, ctypes.byref(P), _hp(rows), u64(rows.size), u64(chunk_cost)))
        else:
            _check(lib().mi_dbg_host_chelpers_run(self.h, ctypes.byref(P), _hp(rows), u64(rows.size)))

    def run_host(self, pols, const_pols, n_const, challenges, publics, x, x_stride, zhinv, q, rows):
        """The same translated program on the CPU (test hook): every array is a host numpy uint64 array."""
        keep = []
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        P = self._params(pols.ctypes.data, const_pols.ctypes.data, n_const, challenges, publics, x.ctypes.data, x_stride, zhinv, q.ctypes.data, keep)
        _check(lib().mi_dbg_host_chelpers_run(self.h, ctypes.byref(P), _hp(rows), u64(rows.size)))


class MultiTree:
    """A tree committed by Multi.commit: rows and subtrees live on the shards."""

    def __init__(self, multi, handle, root, n_ext, ncols):
        self.multi, self.h, self.root, self.n_ext, self.ncols = multi, handle, root, n_ext, ncols
        info = (u64 * 6)()
        _check(lib().mi_multi_tree_info(self.h, info))
        self.shards, self.rows_per_shard, self.cols_per_shard, self.rounds = (int(info[i]) for i in range(4))

    def group_proofs(self, idx, with_values=True):
        """-> [len(idx), ncols + 4 log2(n_ext)] host array: MerkleTreeGL::getGroupProof for every row of idx"""
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        stride = self.ncols + 4 * (self.n_ext - 1).bit_length()
        out = np.zeros((idx.size, stride), dtype=np.uint64)
        _check(lib().mi_multi_group_proofs(self.h, _hp(out), _hp(idx), u64(idx.size), ctypes.c_int(int(with_values))))
        return out

    def gather_rows(self, row0, nrows):
        out = np.empty((nrows, self.ncols), dtype=np.uint64)
        _check(lib().mi_multi_gather_rows(self.h, _hp(out), u64(row0), u64(nrows)))
        return out

    def leaf_digests(self, shard):
        """level-0 digests of the shard's rows, as a host array [rows_per_shard, 4] (for checks)"""
        L = lib()
        L.mi_multi_tree_nodes.restype = ctypes.c_void_p
        L.mi_multi_ctx.restype = ctypes.c_void_p
        out = np.empty((self.rows_per_shard, 4), dtype=np.uint64)
        _check(L.mi_copy_d2h(ctypes.c_void_p(L.mi_multi_ctx(self.multi.h, ctypes.c_int(shard))), _hp(out),
                             ctypes.c_void_p(L.mi_multi_tree_nodes(self.h, ctypes.c_int(shard))), u64(out.size * 8)))
        return out

    def release_rows(self):
        _check(lib().mi_multi_tree_release_rows(self.h))

    def free(self):
        if self.h:
            lib().mi_multi_tree_free(self.h)
            self.h = ctypes.c_void_p()


class Multi:
    """One process, several devices (csrc/multi.hip): the stage commit sharded over `devices` (a device may be named more than once:
    logical shards on one GPU)."""

    def __init__(self, devices, group_same_device=False):
        devs = (ctypes.c_int * len(devices))(*devices)
        self.h = ctypes.c_void_p()
        self.devices = list(devices)
        _check(lib().mi_multi_create2(ctypes.byref(self.h), devs, ctypes.c_int(len(devices)), ctypes.c_int(int(group_same_device))))

    def set_transient(self, on=True):
        """the next commit keeps no rows: a row image for every shard (set_row_images), tiles written once into the images and absorbed there"""
        _check(lib().mi_multi_set_transient(self.h, ctypes.c_int(int(on))))

    @staticmethod
    def transient_need(n, n_ext, ncols, shards):
        """elements of device memory a transient commit takes per device group (what to lend its leader)"""
        lib().mi_multi_transient_need.restype = ctypes.c_uint64
        return int(lib().mi_multi_transient_need(u64(n), u64(n_ext), u64(ncols), ctypes.c_uint32(shards)))

    def ctx_handle(self, shard):
        lib().mi_multi_ctx.restype = ctypes.c_void_p
        return ctypes.c_void_p(lib().mi_multi_ctx(self.h, ctypes.c_int(shard)))

    def commit(self, src_ptr, n, n_ext, ncols, src_device=-1, src_pitch=None, image_ptr=None, image_pitch=None, base_ptr=None, base_pitch=None, image_device=0):
        """src_ptr: address of the n x ncols row-major section (host memory when src_device < 0, else on that device) -> MultiTree"""
        t = ctypes.c_void_p()
        root = np.zeros(4, dtype=np.uint64)
        _check(lib().mi_multi_commit(self.h, ctypes.byref(t), ctypes.c_void_p(src_ptr), u64(src_pitch or ncols), ctypes.c_int(src_device), u64(n), u64(n_ext), u64(ncols),
                                     ctypes.c_void_p(image_ptr), u64(image_pitch or ncols), ctypes.c_void_p(base_ptr), u64(base_pitch or ncols), ctypes.c_int(image_device),
                                     _hp(root)))
        return MultiTree(self, t, root, n_ext, ncols)

    def lend(self, shard, ptr, nbytes):
        """the next commit carves shard `shard`'s row buffers, staging and NTT workspace out of [ptr, ptr + nbytes) (device memory of that shard)"""
        _check(lib().mi_multi_lend(self.h, ctypes.c_int(shard), ctypes.c_void_p(ptr), u64(nbytes)))

    def last_stats(self):
        G = len(self.devices)
        buf = np.zeros(G * (4 + G), dtype=np.float64)
        wall = ctypes.c_double()
        _check(lib().mi_multi_last_stats(self.h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(wall)))
        rows = buf.reshape(G, 4 + G)
        return {"wall_ms": wall.value,
                "per_shard": [{"shard": g, "device": self.devices[g], "lde_ms": float(r[0]), "absorb_ms": float(r[1]), "exchange_wait_ms": float(r[2]),
                               "host_pack_ms": float(r[3]), "bytes_sent_to_shard": [int(b) for b in r[4:]]} for g, r in enumerate(rows)]}

    def peer_access(self):
        """what the driver answered about direct access between the shards' devices: matrix[a][b] = 2 same device, 1 enabled, 0 not
        possible, -1 enabling failed; `indirect_pairs` of them are staged through the host, `warning` says which"""
        G = len(self.devices)
        mat = (ctypes.c_int * (G * G))()
        warn = ctypes.create_string_buffer(2048)
        bad = int(lib().mi_multi_peer_access(self.h, mat, warn, u64(2048)))
        return {"matrix": [[int(mat[a * G + b]) for b in range(G)] for a in range(G)], "indirect_pairs": bad, "warning": warn.value.decode()}

    def set_pack_threads(self, threads):
        _check(lib().mi_multi_set_pack_threads(self.h, ctypes.c_int(threads)))

    def set_row_images(self, ptrs, pitch, halo_rows):
        """the next commit also leaves every shard's own rows (+ halo_rows after them, wrapping) row-major in ptrs[shard] (0 / None: not for that shard)"""
        arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(p or None) for p in ptrs])
        _check(lib().mi_multi_set_row_images(self.h, arr, u64(pitch), u64(halo_rows)))

    def set_upload_mode(self, mode):
        """-1 auto (page-locked host source -> strided DMA per device link, pageable -> host-packed staging), 0 packed, 1 strided"""
        _check(lib().mi_multi_set_upload_mode(self.h, ctypes.c_int(mode)))

    def last_upload_mode(self):
        return {-1: "device", 0: "packed", 1: "strided"}[int(lib().mi_multi_last_upload_mode(self.h))]

    def close(self):
        if self.h:
            lib().mi_multi_destroy(self.h)
            self.h = ctypes.c_void_p()


def multi_check_stats():
    """MI_MULTI_CHECK=1: {enabled, checks, unknown (pointers nobody entered: they pass), violations}"""
    out = (ctypes.c_uint64 * 4)()
    _check(lib().mi_multi_check_stats(out))
    return {"enabled": bool(out[0]), "checks": int(out[1]), "unknown": int(out[2]), "violations": int(out[3])}


# ---- host debug hooks (same inline math as the kernels, run on the CPU; tests only)
def dbg_host_permute(state, variant):
    s = np.ascontiguousarray(state, dtype=np.uint64).copy()
    lib().mi_dbg_host_poseidon_permute(_hp(s), ctypes.c_int(variant))
    return s


def dbg_host_mul(a, b):
    return int(lib().mi_dbg_host_mul(u64(a), u64(b)))


def dbg_host_e3_mul(a, b):
    out = np.empty(3, dtype=np.uint64)
    lib().mi_dbg_host_e3_mul(_hp(out), _hp(np.ascontiguousarray(a, dtype=np.uint64)), _hp(np.ascontiguousarray(b, dtype=np.uint64)))
    return out


def dbg_host_e3_inv(a):
    out = np.empty(3, dtype=np.uint64)
    lib().mi_dbg_host_e3_inv(_hp(out), _hp(np.ascontiguousarray(a, dtype=np.uint64)))
    return out


def dbg_host_dft(x, log_size, inverse):
    v = np.zeros(16, dtype=np.uint64)
    v[:1 << log_size] = x
    lib().mi_dbg_host_dft16(_hp(v), ctypes.c_int(log_size), ctypes.c_int(int(inverse)))
    return v[:1 << log_size].copy()
