"""ctypes binding of the C ABI in include/porefv.h.

The product library is ``porepy_amd/csrc/libporefv_hip.so`` (hipcc, gfx950).  There is no
CPU fallback: if the library is missing, was not built for the device, or no GPU is
visible, creating a context raises.  ``load_library(path)`` exists so that tests can bind
the host-emulation build of the same sources explicitly; nothing in this package does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIBRARY = os.path.join(_HERE, "csrc", "libporefv_hip.so")

MAT_FLUX, MAT_BOUND_FLUX, MAT_BOUND_PRESSURE_CELL, MAT_BOUND_PRESSURE_FACE = 0, 1, 2, 3
MAT_VECTOR_SOURCE, MAT_BOUND_PRESSURE_VECTOR_SOURCE, MAT_SYSTEM = 4, 5, 6
MAT_STRESS, MAT_BOUND_STRESS, MAT_BOUND_DISPLACEMENT_CELL, MAT_BOUND_DISPLACEMENT_FACE, MAT_MECH_SYSTEM = 7, 8, 9, 10, 11
MAT_USER_SYSTEM = 12
MAT_FLUX_JACOBIAN = 13
BC_DIR, BC_NEU, BC_ROB, BC_INTERNAL = 1, 2, 4, 8
SOLVE_CG, SOLVE_BICGSTAB, SOLVE_GMRES = 0, 1, 2
DISCR_REBUILD_TOPOLOGY, DISCR_SKIP_VECTOR_SOURCE = 1, 2

STATUS_NAMES = {
    0: "ok", 1: "singular local system", 2: "unsupported cell shape", 3: "HIP error",
    4: "bad argument", 5: "unsupported size/feature", 6: "not converged",
}

EXPORTS = [
    "pfv_create", "pfv_destroy", "pfv_last_error", "pfv_is_device_build", "pfv_set_grid",
    "pfv_mpfa_set_params", "pfv_mpfa_discretize", "pfv_matrix_info", "pfv_get_matrix",
    "pfv_mpfa_assemble", "pfv_get_rhs", "pfv_spmv", "pfv_solve", "pfv_spmv_device",
    "pfv_get_device_rhs", "pfv_sync", "pfv_get_stats", "pfv_time_kernel", "pfv_debug_copy",
    "pfv_spmv_device_rows", "pfv_copy_device_vector", "pfv_set_stream",
    "pfv_mpsa_set_params", "pfv_mpsa_discretize", "pfv_mpsa_assemble",
    "pfv_mpfa_discretize_faces", "pfv_set_system", "pfv_tpfa_discretize", "pfv_mpsa_discretize_faces", "pfv_set_preconditioner", "pfv_amg_setup", "pfv_amg_setup_sharded", "pfv_amg_apply_device", "pfv_csr_from_host", "pfv_csr_from_matrix", "pfv_csr_block_diag", "pfv_csr_matmul", "pfv_csr_axpby", "pfv_csr_transpose", "pfv_csr_bmat", "pfv_csr_scale", "pfv_csr_divide", "pfv_csr_spmv", "pfv_csr_spmv_device", "pfv_csr_info", "pfv_csr_get", "pfv_csr_set_system", "pfv_csr_free", "pfv_reset_stream", "pfv_mpsa_set_robin", "pfv_mpsa_set_basis", "pfv_mpfa_set_subface_bc", "pfv_mpsa_set_subface_bc", "pfv_mpsa_set_subface_basis",
    "pfv_biot_set_alphas", "pfv_biot_discretize", "pfv_biot_matrix_info", "pfv_biot_get_matrix",
    "pfv_set_vectors_on_device", "pfv_set_periodic", "pfv_biot_discretize_faces", "pfv_solve_sharded", "pfv_tpfa_transmissibility_ad",
    "pfv_get_matrix_rows", "pfv_active_size", "pfv_device_memory",
    "pfv_rccl_unique_id", "pfv_rccl_comm_create", "pfv_rccl_set_halo_plan", "pfv_rccl_hooks", "pfv_rccl_stats",
    "pfv_rccl_last_error", "pfv_rccl_comm_destroy", "pfv_mpfa_ad_flux_system", "pfv_host_alloc", "pfv_host_free",
    "pfv_mpsa_set_subface_eta", "pfv_mpsa_set_reconstruction_eta", "pfv_mpsa_set_reconstruction_eta_subface", "pfv_get_stats_n", "pfv_set_block_preconditioner",
    "pfv_mpfa_set_permeability",
]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("rel_residual", C.c_double), ("solve_ms", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("topology_ms", C.c_double), ("symbolic_ms", C.c_double), ("node_ms", C.c_double),
                ("face_ms", C.c_double), ("assemble_ms", C.c_double), ("solve_ms", C.c_double),
                ("bytes_written_outputs", C.c_double), ("num_nodes", C.c_int64),
                ("num_sub_half_faces", C.c_int64), ("sum_block_sq", C.c_int64),
                ("max_block", C.c_int64), ("amg_setup_ms", C.c_double),
                ("amg_operator_complexity", C.c_double), ("amg_levels", C.c_int64),
                ("amg_coarsest_rows", C.c_int64), ("discretize_ms", C.c_double), ("solve_renumbered", C.c_int64),
                ("node_flops", C.c_double), ("node_table_doubles", C.c_int64),
                ("amg_maps_reused", C.c_int64), ("amg_level0_nnz", C.c_int64), ("amg_filter_theta", C.c_double),
                ("win_reused", C.c_int64), ("amg_filter_layout", C.c_int64),
                ("solve_launches", C.c_int64), ("amg_setup_launches", C.c_int64), ("node_redo", C.c_int64),
                ("symbolic_reused", C.c_int64), ("amg_stale_rematches", C.c_int64),
                ("mpsa_contrast_regions", C.c_int64), ("mpsa_max_contrast", C.c_double), ("assemble_positions_kept", C.c_int64),
                ("pipeline_runs", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)            # (user, d_x, stream)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)  # (user, d_vals, count, stream)


# (user, n_peers, peers, d_send, send_ptr, d_recv, recv_ptr, stream) / (user, d_send, d_recv, bytes_per_rank, stream)
SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int64),
                          C.c_void_p, C.POINTER(C.c_int64), C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class ShardHooks(C.Structure):
    """pfv_shard_hooks of include/porefv.h."""
    _fields_ = [("exchange_halo", HALO_FN), ("allreduce_sum", ALLREDUCE_FN), ("user", C.c_void_p),
                ("sendrecv", SENDRECV_FN), ("allgather", ALLGATHER_FN)]


class PorefvError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"porefv status {status} ({STATUS_NAMES.get(status, '?')}): {message}")
        self.status = status
        self.message = message


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_int8)
_up = C.POINTER(C.c_uint8)
_lp = C.POINTER(C.c_int64)
_h = C.c_void_p


def _bind(lib: C.CDLL) -> C.CDLL:
    lib.pfv_create.argtypes = [C.c_int, C.POINTER(_h)]
    lib.pfv_create.restype = C.c_int
    lib.pfv_destroy.argtypes = [_h]
    lib.pfv_destroy.restype = None
    lib.pfv_last_error.argtypes = [_h]
    lib.pfv_last_error.restype = C.c_char_p
    lib.pfv_is_device_build.argtypes = []
    lib.pfv_is_device_build.restype = C.c_int
    lib.pfv_set_grid.argtypes = [_h, C.c_int, C.c_int64, C.c_int64, C.c_int64, _dp, _ip, _ip, _bp,
                                 _ip, _ip, _dp, _dp, _dp, _dp]
    lib.pfv_set_grid.restype = C.c_int
    lib.pfv_mpfa_set_params.argtypes = [_h, _dp, _up, _dp, C.c_double, _dp]
    lib.pfv_mpfa_set_params.restype = C.c_int
    lib.pfv_mpfa_set_permeability.argtypes = [_h, _dp]
    lib.pfv_mpfa_set_permeability.restype = C.c_int
    lib.pfv_mpfa_discretize.argtypes = [_h, C.c_uint32]
    lib.pfv_mpfa_discretize.restype = C.c_int
    lib.pfv_mpfa_discretize_faces.argtypes = [_h, C.c_uint32, C.c_int64, _ip, C.c_int]
    lib.pfv_mpfa_discretize_faces.restype = C.c_int
    lib.pfv_mpsa_discretize_faces.argtypes = [_h, C.c_uint32, C.c_int64, _ip, C.c_int]
    lib.pfv_mpsa_discretize_faces.restype = C.c_int
    lib.pfv_biot_set_alphas.argtypes = [_h, C.c_int, _dp]
    lib.pfv_biot_set_alphas.restype = C.c_int
    lib.pfv_biot_discretize.argtypes = [_h, C.c_uint32]
    lib.pfv_biot_discretize.restype = C.c_int
    lib.pfv_biot_matrix_info.argtypes = [_h, C.c_int, _lp, _lp, _lp]
    lib.pfv_biot_matrix_info.restype = C.c_int
    lib.pfv_biot_get_matrix.argtypes = [_h, C.c_int, C.c_int, _ip, _ip, _dp]
    lib.pfv_biot_get_matrix.restype = C.c_int
    lib.pfv_mpfa_set_subface_bc.argtypes = [_h, _up, _dp]
    lib.pfv_mpfa_set_subface_bc.restype = C.c_int
    lib.pfv_mpsa_set_basis.argtypes = [_h, _dp]
    lib.pfv_mpsa_set_basis.restype = C.c_int
    lib.pfv_mpsa_set_subface_bc.argtypes = [_h, _up, _up, _up, _dp]
    lib.pfv_mpsa_set_subface_bc.restype = C.c_int
    lib.pfv_mpsa_set_subface_basis.argtypes = [_h, _dp]
    lib.pfv_mpsa_set_subface_basis.restype = C.c_int
    lib.pfv_mpsa_set_robin.argtypes = [_h, _up, _dp]
    lib.pfv_mpsa_set_robin.restype = C.c_int
    lib.pfv_mpsa_set_subface_eta.argtypes = [_h, _dp]
    lib.pfv_mpsa_set_subface_eta.restype = C.c_int
    lib.pfv_mpsa_set_reconstruction_eta.argtypes = [_h, C.c_int, C.c_double]
    lib.pfv_mpsa_set_reconstruction_eta.restype = C.c_int
    lib.pfv_mpsa_set_reconstruction_eta_subface.argtypes = [_h, _dp]
    lib.pfv_mpsa_set_reconstruction_eta_subface.restype = C.c_int
    lib.pfv_reset_stream.argtypes = [_h]
    lib.pfv_reset_stream.restype = C.c_int
    lib.pfv_amg_setup.argtypes = [_h, C.c_int64]
    lib.pfv_amg_setup.restype = C.c_int
    lib.pfv_amg_setup_sharded.argtypes = [_h, C.c_int64, C.POINTER(ShardHooks), C.c_int, C.c_int, C.c_int, _ip, _lp, _ip,
                                          _lp, _ip]
    lib.pfv_amg_setup_sharded.restype = C.c_int
    # device-resident CSR algebra (csr_algebra.inc)
    lib.pfv_csr_from_host.argtypes = [_h, C.c_int64, C.c_int64, _ip, _ip, _dp, C.POINTER(_h)]
    lib.pfv_csr_from_matrix.argtypes = [_h, _h, C.c_int, C.POINTER(_h)]
    lib.pfv_csr_block_diag.argtypes = [_h, C.c_int, C.POINTER(_h), C.POINTER(_h)]
    lib.pfv_csr_matmul.argtypes = [_h, _h, _h, C.POINTER(_h)]
    lib.pfv_csr_axpby.argtypes = [_h, C.c_double, _h, C.c_double, _h, C.POINTER(_h)]
    lib.pfv_csr_transpose.argtypes = [_h, _h, C.POINTER(_h)]
    lib.pfv_csr_bmat.argtypes = [_h, C.c_int, C.c_int, C.POINTER(_h), _lp, _lp, C.POINTER(_h)]
    lib.pfv_csr_scale.argtypes = [_h, _dp, _dp]
    lib.pfv_csr_divide.argtypes = [_h, C.c_double]
    lib.pfv_csr_divide.restype = C.c_int
    lib.pfv_csr_spmv.argtypes = [_h, _dp, _dp]
    lib.pfv_csr_spmv_device.argtypes = [_h, C.c_void_p, C.c_void_p]
    lib.pfv_csr_info.argtypes = [_h, _lp, _lp, _lp]
    lib.pfv_csr_get.argtypes = [_h, _ip, _ip, _dp]
    lib.pfv_csr_set_system.argtypes = [_h, _h, C.c_void_p, C.c_int]
    for name in ("from_host", "from_matrix", "block_diag", "matmul", "axpby", "transpose", "bmat", "scale", "spmv", "spmv_device", "info", "get",
                 "set_system"):
        getattr(lib, "pfv_csr_" + name).restype = C.c_int
    lib.pfv_csr_free.argtypes = [_h]
    lib.pfv_csr_free.restype = None
    lib.pfv_amg_apply_device.argtypes = [_h, C.c_void_p, C.c_void_p]
    lib.pfv_amg_apply_device.restype = C.c_int
    lib.pfv_biot_discretize_faces.argtypes = [_h, C.c_uint32, C.c_int64, _ip, C.c_int64, _ip, C.c_int]
    lib.pfv_biot_discretize_faces.restype = C.c_int
    lib.pfv_set_periodic.argtypes = [_h, C.POINTER(C.c_int32), _dp]
    lib.pfv_set_periodic.restype = C.c_int
    lib.pfv_set_vectors_on_device.argtypes = [_h, C.c_int]
    lib.pfv_set_vectors_on_device.restype = C.c_int
    lib.pfv_set_preconditioner.argtypes = [_h, C.c_int]
    lib.pfv_set_preconditioner.restype = C.c_int
    lib.pfv_tpfa_discretize.argtypes = [_h, C.c_int]
    lib.pfv_tpfa_discretize.restype = C.c_int
    lib.pfv_set_system.argtypes = [_h, C.c_int64, _ip, _ip, _dp, _dp]
    lib.pfv_set_system.restype = C.c_int
    lib.pfv_matrix_info.argtypes = [_h, C.c_int, _lp, _lp, _lp]
    lib.pfv_matrix_info.restype = C.c_int
    lib.pfv_get_matrix.argtypes = [_h, C.c_int, _ip, _ip, _dp]
    lib.pfv_get_matrix.restype = C.c_int
    lib.pfv_mpfa_assemble.argtypes = [_h, _dp, _dp, _dp]
    lib.pfv_mpfa_assemble.restype = C.c_int
    lib.pfv_get_rhs.argtypes = [_h, _dp]
    lib.pfv_get_rhs.restype = C.c_int
    lib.pfv_spmv.argtypes = [_h, C.c_int, _dp, _dp]
    lib.pfv_spmv.restype = C.c_int
    lib.pfv_solve.argtypes = [_h, C.c_int, C.c_double, C.c_int, C.c_int, _dp, _dp, C.POINTER(SolveInfo)]
    lib.pfv_solve.restype = C.c_int
    lib.pfv_spmv_device.argtypes = [_h, C.c_int, C.c_void_p, C.c_void_p]
    lib.pfv_spmv_device.restype = C.c_int
    lib.pfv_get_device_rhs.argtypes = [_h, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.pfv_get_device_rhs.restype = C.c_int
    lib.pfv_sync.argtypes = [_h]
    lib.pfv_sync.restype = C.c_int
    lib.pfv_get_stats.argtypes = [_h, C.POINTER(Stats)]
    lib.pfv_get_stats.restype = C.c_int
    lib.pfv_set_block_preconditioner.argtypes = [_h, C.c_int64, C.POINTER(C.c_int64), C.c_int]
    lib.pfv_set_block_preconditioner.restype = C.c_int
    lib.pfv_get_stats_n.argtypes = [_h, C.c_void_p, C.c_size_t]
    lib.pfv_get_stats_n.restype = C.c_int
    lib.pfv_time_kernel.argtypes = [_h, C.c_int, C.c_int, _dp]
    lib.pfv_time_kernel.restype = C.c_int
    lib.pfv_mpfa_ad_flux_system.argtypes = [_h, _dp, _dp, _dp, _dp, _dp, _dp, C.c_uint32]
    lib.pfv_mpfa_ad_flux_system.restype = C.c_int
    _i64p = C.POINTER(C.c_int64)
    lib.pfv_rccl_unique_id.argtypes = [C.c_char_p]
    lib.pfv_rccl_unique_id.restype = C.c_int
    lib.pfv_rccl_comm_create.argtypes = [_h, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.pfv_rccl_comm_create.restype = C.c_int
    lib.pfv_rccl_set_halo_plan.argtypes = [C.c_void_p, C.c_int, _ip, _i64p, _ip, _i64p, _ip]
    lib.pfv_rccl_set_halo_plan.restype = C.c_int
    lib.pfv_rccl_hooks.argtypes = [C.c_void_p, C.POINTER(ShardHooks)]
    lib.pfv_rccl_hooks.restype = C.c_int
    lib.pfv_rccl_stats.argtypes = [C.c_void_p, _i64p, _i64p, _i64p]
    lib.pfv_rccl_stats.restype = C.c_int
    lib.pfv_rccl_last_error.argtypes = [C.c_void_p]
    lib.pfv_rccl_last_error.restype = C.c_char_p
    lib.pfv_rccl_comm_destroy.argtypes = [C.c_void_p]
    lib.pfv_rccl_comm_destroy.restype = None
    lib.pfv_device_memory.argtypes = [_h, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.pfv_device_memory.restype = C.c_int
    lib.pfv_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    lib.pfv_host_alloc.restype = C.c_int
    lib.pfv_host_free.argtypes = [C.c_void_p]
    lib.pfv_host_free.restype = None
    lib.pfv_active_size.argtypes = [_h, C.POINTER(C.c_int64)]
    lib.pfv_active_size.restype = C.c_int
    lib.pfv_get_matrix_rows.argtypes = [_h, C.c_int, C.c_int64, _ip, _ip, _ip, _dp]
    lib.pfv_get_matrix_rows.restype = C.c_int
    lib.pfv_debug_copy.argtypes = [_h, C.c_int, _dp, C.c_int64]
    lib.pfv_debug_copy.restype = C.c_int
    lib.pfv_spmv_device_rows.argtypes = [_h, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    lib.pfv_spmv_device_rows.restype = C.c_int
    lib.pfv_copy_device_vector.argtypes = [_h, C.c_int, C.c_void_p, C.c_int64]
    lib.pfv_copy_device_vector.restype = C.c_int
    lib.pfv_tpfa_transmissibility_ad.argtypes = [_h, _dp, _dp, _dp]
    lib.pfv_tpfa_transmissibility_ad.restype = C.c_int
    lib.pfv_solve_sharded.argtypes = [_h, C.c_int, C.c_double, C.c_int, C.c_int64, C.POINTER(ShardHooks),
                                      C.c_void_p, C.c_void_p, C.POINTER(SolveInfo)]
    lib.pfv_solve_sharded.restype = C.c_int
    lib.pfv_set_stream.argtypes = [_h, C.c_void_p]
    lib.pfv_set_stream.restype = C.c_int
    lib.pfv_mpsa_set_params.argtypes = [_h, _dp, _dp, _up, _up, C.c_double]
    lib.pfv_mpsa_set_params.restype = C.c_int
    lib.pfv_mpsa_discretize.argtypes = [_h, C.c_uint32]
    lib.pfv_mpsa_discretize.restype = C.c_int
    lib.pfv_mpsa_assemble.argtypes = [_h, _dp, _dp]
    lib.pfv_mpsa_assemble.restype = C.c_int
    return lib


def load_library(path: str) -> C.CDLL:
    """Bind a build of the porefv sources at an explicit path (tests use this for the
    host-emulation build; the product only ever binds DEFAULT_LIBRARY)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    return _bind(C.CDLL(path))


_product = None


def product_library() -> C.CDLL:
    """The gfx950 HIP library.  Fails loudly when it is missing or is not a device build."""
    global _product
    if _product is None:
        if not os.path.exists(DEFAULT_LIBRARY):
            raise RuntimeError(
                f"{DEFAULT_LIBRARY} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        lib = _bind(C.CDLL(DEFAULT_LIBRARY))
        if not lib.pfv_is_device_build():
            raise RuntimeError(f"{DEFAULT_LIBRARY} is not a HIP device build")
        _product = lib
    return _product


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def rccl_unique_id(library=None) -> bytes:
    """128-byte id of a new RCCL communicator (rank 0 calls this and distributes the bytes)."""
    lib = library if library is not None else product_library()
    buf = C.create_string_buffer(128)
    st = lib.pfv_rccl_unique_id(buf)
    if st != 0:
        raise PorefvError(st, "RCCL is not available (gfx950 build with librccl.so needed)")
    return buf.raw


def rccl_available(library=None) -> bool:
    """Can this process load librccl through the library?  A purely local probe (ncclGetUniqueId does not
    communicate): ranks ask it before the collective communicator initialisation."""
    try:
        rccl_unique_id(library)
        return True
    except PorefvError:
        return False


class RcclComm:
    """Native RCCL transport of a sharded solve (include/porefv.h: pfv_rccl_*): one communicator per rank on
    the device of ``ctx``, plus this rank's halo plan.  Pass it to ``Context.solve_sharded`` in place of the
    two Python callables."""

    def __init__(self, ctx: "Context", unique_id: bytes, rank: int, world: int):
        self.ctx = ctx
        self.lib = ctx.lib
        self._c = C.c_void_p()
        if not isinstance(unique_id, (bytes, bytearray)) or len(unique_id) != 128:
            raise PorefvError(2, "RCCL unique id must be the 128 bytes of rccl_unique_id()")  # (the library reads 128)
        st = self.lib.pfv_rccl_comm_create(ctx._h, bytes(unique_id), int(rank), int(world), C.byref(self._c))
        if st != 0:
            raise PorefvError(st, self.lib.pfv_last_error(ctx._h).decode(errors="replace"))
        self.rank, self.world = int(rank), int(world)
        import weakref

        ctx._rccl_refs.append(weakref.ref(self))

    def set_halo_plan(self, send: dict, recv: dict):
        """send[p] = local indices of the owned entries peer p needs (in the order it expects them),
        recv[p] = positions of the halo entries peer p owns (in the order it sends them)."""
        peers = sorted(set(send) | set(recv))
        sp, rp, si, ri = [0], [0], [], []
        for p in peers:
            a = np.asarray(send.get(p, []), dtype=np.int32).ravel()
            b = np.asarray(recv.get(p, []), dtype=np.int32).ravel()
            si.append(a)
            ri.append(b)
            sp.append(sp[-1] + a.size)
            rp.append(rp[-1] + b.size)
        pe = np.asarray(peers, dtype=np.int32)
        spa, rpa = np.asarray(sp, dtype=np.int64), np.asarray(rp, dtype=np.int64)
        sia = np.concatenate(si).astype(np.int32) if si else np.zeros(0, np.int32)
        ria = np.concatenate(ri).astype(np.int32) if ri else np.zeros(0, np.int32)
        st = self.lib.pfv_rccl_set_halo_plan(self._c, len(peers), _ptr(pe, _ip), _ptr(spa, C.POINTER(C.c_int64)),
                                             _ptr(sia, _ip), _ptr(rpa, C.POINTER(C.c_int64)), _ptr(ria, _ip))
        if st != 0:
            raise PorefvError(st, self.lib.pfv_last_error(self.ctx._h).decode(errors="replace"))

    def stats(self) -> dict:
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self.lib.pfv_rccl_stats(self._c, C.byref(a), C.byref(b), C.byref(c))
        return {"exchanges": a.value, "allreduces": b.value, "bytes_per_exchange": c.value}

    def last_error(self) -> str:
        return (self.lib.pfv_rccl_last_error(self._c) or b"").decode(errors="replace")

    def close(self):
        if getattr(self, "_c", None) is not None and self._c:
            # The communicator lives on its handle's stream and memory pool.  When both die in one garbage cycle the
            # handle's finalizer may already have run (weak references are cleared before finalizers, so the handle
            # could not close this object first): its memory went with the handle, nothing is left to destroy.
            if getattr(self.ctx, "_h", None):
                self.lib.pfv_rccl_comm_destroy(self._c)
            self._c = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """Owner of one page-locked block: when the last numpy view of it dies the block goes back to its pool."""
    __slots__ = ("pool", "ptr", "nbytes")

    def __init__(self, pool, ptr, nbytes):
        self.pool, self.ptr, self.nbytes = pool, ptr, nbytes

    def __del__(self):
        try:
            self.pool._give(self.ptr, self.nbytes)
        except Exception:
            pass


class PinnedPool:
    """Page-locked host arrays for the results copied out of the device (include/porefv.h: pfv_host_alloc).

    ``empty(n, dtype)`` returns an ordinary numpy array whose memory is a page-locked block; when the array (and
    whatever scipy matrix holds it) is garbage-collected the block is kept for the next request of the same size --
    the matrices of a time-stepping loop have the same sizes step after step, so after the first step no memory is
    pinned, unpinned or page-faulted any more.  Requests below ``min_bytes`` use plain numpy arrays; at most
    ``PFV_PINNED_POOL_GB`` (default: a quarter of the available host memory, at most 48) are kept, and the last ``Context`` of a library to close releases them; ``PFV_PINNED_POOL=0`` switches the pool off."""

    min_bytes = 1 << 20

    def __init__(self, lib):
        import os

        self.lib = lib
        self.free: dict = {}
        self.cached = 0
        # cap of what is KEPT page-locked between uses: PFV_PINNED_POOL_GB, default a quarter of the memory that is
        # available now (at most 48 GB) -- the eager matrices of a 2 M-cell grid are 23 GB; a small host keeps less
        default_gb = 48.0
        try:
            with open("/proc/meminfo") as fh:
                for line in fh:
                    if line.startswith("MemAvailable:"):
                        default_gb = min(48.0, 0.25 * int(line.split()[1]) / (1 << 20))
                        break
        except OSError:
            pass
        self.cap = int(float(os.environ.get("PFV_PINNED_POOL_GB", str(default_gb))) * (1 << 30))
        self.enabled = os.environ.get("PFV_PINNED_POOL", "1") not in ("0", "")
        self.allocated = 0  # statistics: blocks ever page-locked / requests served from the pool
        self.reused = 0

    def empty(self, n: int, dtype) -> np.ndarray:
        dt = np.dtype(dtype)
        nbytes = int(n) * dt.itemsize
        if not self.enabled or nbytes < self.min_bytes:
            return np.empty(int(n), dtype=dt)
        lst = self.free.get(nbytes)
        if lst:
            ptr = lst.pop()
            self.cached -= nbytes
            self.reused += 1
        else:
            p = C.c_void_p()
            if self.lib.pfv_host_alloc(nbytes, C.byref(p)) != 0 or not p.value:
                return np.empty(int(n), dtype=dt)  # no page-locked memory left: pageable
            ptr = p.value
            self.allocated += 1
        buf = (C.c_char * nbytes).from_address(ptr)
        buf._pfv_owner = _PinnedBlock(self, ptr, nbytes)  # dies with the last view of buf
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def _give(self, ptr, nbytes):
        if self.lib is None:
            return
        if self.cached + nbytes <= self.cap:
            self.free.setdefault(nbytes, []).append(ptr)
            self.cached += nbytes
        else:
            self.lib.pfv_host_free(ptr)

    def trim(self):
        for lst in self.free.values():
            for ptr in lst:
                self.lib.pfv_host_free(ptr)
        self.free.clear()
        self.cached = 0


_POOLS: dict = {}
_OPEN_CONTEXTS: dict = {}  # id(library) -> live Context handles


def pinned_pool(lib) -> PinnedPool:
    """The pool of a library (one per loaded library: the product's blocks are hipHostMalloc'ed, the emulation's malloc'ed)."""
    key = id(lib)
    if key not in _POOLS:
        _POOLS[key] = PinnedPool(lib)
    return _POOLS[key]


def free_device_bytes(device: int = 0, library=None) -> int:
    """Free HBM on ``device`` right now, through a handle that lives only for the query (no stream or memory pool is
    kept behind)."""
    ctx = Context(device, library)
    try:
        return ctx.free_device_bytes()
    finally:
        ctx.close()


class Context:
    """One device handle (one GPU, one HIP stream): grid + parameters + results in HBM."""

    def __init__(self, device: int = 0, library: C.CDLL | None = None):
        self.lib = library if library is not None else product_library()
        self._h = _h()
        st = self.lib.pfv_create(int(device), C.byref(self._h))
        if st != 0:
            self._h = _h()
            raise PorefvError(st, "pfv_create failed: no usable MI355X / HIP device "
                                  "(the product path has no CPU fallback)")
        _OPEN_CONTEXTS[id(self.lib)] = _OPEN_CONTEXTS.get(id(self.lib), 0) + 1
        self.nd = self.nc = self.nf = self.nn = 0
        self._discretized = False  # a complete MPFA discretization is resident on the device
        self._discretized_m = False  # ... MPSA
        self._discretized_b = False  # ... Biot coupling terms
        self._lazy_refs: list = []   # weak references to LazyCsr proxies of this handle's matrices (lazy.py)
        self._rccl_refs: list = []   # weak references to the RcclComm objects made on this handle (closed before it)
        self._csr_refs: list = []    # ... and to the DeviceCsr matrices it owns (device_csr.py)

    def _before_overwrite(self):
        """Matrices are about to be recomputed: proxies of the current ones fetch their values first."""
        if self._lazy_refs:
            from .lazy import detach_all

            detach_all(self)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            # communicators hold the handle's stream and memory pool: they go first (a communicator destroyed after
            # its handle would touch freed memory)
            for ref in getattr(self, "_rccl_refs", []):
                comm = ref()
                if comm is not None:
                    comm.close()
            for ref in getattr(self, "_csr_refs", []):
                m = ref()
                if m is not None:
                    m.close()
            self.lib.pfv_destroy(self._h)
            self._h = _h()
            # the last handle of a library to go releases the page-locked blocks its result arrays were recycled through
            n_open = _OPEN_CONTEXTS.get(id(self.lib), 1) - 1
            _OPEN_CONTEXTS[id(self.lib)] = n_open
            if n_open <= 0 and id(self.lib) in _POOLS:
                _POOLS[id(self.lib)].trim()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int):
        if st != 0:
            raise PorefvError(st, self.lib.pfv_last_error(self._h).decode(errors="replace"))

    # ---- inputs -------------------------------------------------------------------
    def set_grid(self, raw: dict):
        self._before_overwrite()
        self._discretized = False
        self._discretized_m = False
        self._discretized_b = False
        nd = int(raw["dim"])
        nodes = _f64(raw["nodes"]); fn_ = _f64(raw["face_normals"]); fc = _f64(raw["face_centers"])
        cc = _f64(raw["cell_centers"]); fa = _f64(raw["face_areas"])
        cfp = np.ascontiguousarray(raw["cf_indptr"], dtype=np.int32)
        cfi = np.ascontiguousarray(raw["cf_indices"], dtype=np.int32)
        cfs = np.ascontiguousarray(raw["cf_sign"], dtype=np.int8)
        fnp = np.ascontiguousarray(raw["fn_indptr"], dtype=np.int32)
        fni = np.ascontiguousarray(raw["fn_indices"], dtype=np.int32)
        nc, nf, nn = cc.shape[1], fc.shape[1], nodes.shape[1]
        if cfp.size != nc + 1 or fnp.size != nf + 1 or nodes.shape[0] != 3:
            raise ValueError("inconsistent grid arrays")
        self._check(self.lib.pfv_set_grid(
            self._h, nd, nc, nf, nn, _ptr(nodes, _dp), _ptr(cfp, _ip), _ptr(cfi, _ip), _ptr(cfs, _bp),
            _ptr(fnp, _ip), _ptr(fni, _ip), _ptr(fn_, _dp), _ptr(fc, _dp), _ptr(cc, _dp), _ptr(fa, _dp)))
        self.nd, self.nc, self.nf, self.nn = nd, nc, nf, nn
        self.nsf = int(fnp[-1])
        self.ncf = int(cfp[-1])

    def set_params(self, perm, bc_flags, robin_weight=None, eta=0.0, eta_subface=None):
        perm = _f64(perm)
        if perm.shape != (3, 3, self.nc):
            raise ValueError(f"permeability must have shape (3, 3, {self.nc})")
        flags = np.ascontiguousarray(bc_flags, dtype=np.uint8)
        if flags.shape != (self.nf,):
            raise ValueError("bc flags must have one entry per face")
        rw = None if robin_weight is None else _f64(robin_weight)
        es = None if eta_subface is None else _f64(eta_subface)
        if es is not None and es.shape != (self.nsf,):
            raise ValueError("size of eta must either be 1 or number of subfaces")
        self._check(self.lib.pfv_mpfa_set_params(self._h, _ptr(perm, _dp), _ptr(flags, _up),
                                                 _ptr(rw, _dp), float(eta), _ptr(es, _dp)))

    def set_permeability(self, perm):
        """New permeability values (3, 3, Nc), everything else of ``set_params`` kept."""
        perm = _f64(perm)
        if perm.shape != (3, 3, self.nc):
            raise ValueError(f"permeability must have shape (3, 3, {self.nc})")
        self._check(self.lib.pfv_mpfa_set_permeability(self._h, _ptr(perm, _dp)))

    def set_permeability_device(self, perm_ptr: int):
        """``set_permeability`` from a device buffer of 9 Nc doubles ((3, 3, Nc) C-order), copied device-to-device."""
        self._dev(True)
        try:
            self._check(self.lib.pfv_mpfa_set_permeability(self._h, C.cast(perm_ptr, _dp)))
        finally:
            self._dev(False)

    def set_subface_bc(self, bc_flags_sub, robin_weight_sub=None):
        """Boundary conditions per sub-face (face_nodes CSC order, sorted indices); None switches back to
        the per-face conditions of set_params."""
        if bc_flags_sub is None:
            self._check(self.lib.pfv_mpfa_set_subface_bc(self._h, None, None))
            return
        fl = np.ascontiguousarray(bc_flags_sub, dtype=np.uint8)
        if fl.shape != (self.nsf,):
            raise ValueError("one flag per sub-face expected")
        rw = None if robin_weight_sub is None else _f64(robin_weight_sub)
        if rw is not None and rw.shape != (self.nsf,):
            raise ValueError("one Robin weight per sub-face expected")
        self._check(self.lib.pfv_mpfa_set_subface_bc(self._h, _ptr(fl, _up), _ptr(rw, _dp)))
        self._discretized = False

    # ---- MPSA ----------------------------------------------------------------------
    def mpsa_set_params(self, stiffness, cell_volumes, is_dir, is_neu, eta=0.0, is_rob=None, robin_weight=None,
                        basis=None):
        C9 = _f64(stiffness)
        if C9.shape != (9, 9, self.nc):
            raise ValueError(f"stiffness must have shape (9, 9, {self.nc})")
        vol = _f64(cell_volumes)
        is_dir, is_neu = np.asarray(is_dir, bool), np.asarray(is_neu, bool)
        if is_dir.shape != (self.nd, self.nf) or is_neu.shape != (self.nd, self.nf):
            raise AttributeError("MPSA needs a vectorial boundary condition: is_dir / is_neu of shape (nd, Nf)")
        wts = (1 << np.arange(self.nd))[:, None]
        dbits = np.ascontiguousarray((is_dir * wts).sum(axis=0), dtype=np.uint8)
        nbits = np.ascontiguousarray((is_neu * wts).sum(axis=0), dtype=np.uint8)
        self._check(self.lib.pfv_mpsa_set_params(self._h, _ptr(C9, _dp), _ptr(vol, _dp), _ptr(dbits, _up),
                                                 _ptr(nbits, _up), float(eta)))
        if is_rob is not None and np.any(is_rob):
            is_rob = np.asarray(is_rob, bool)
            if is_rob.shape != (self.nd, self.nf):
                raise AttributeError("is_rob must have shape (nd, Nf)")
            rbits = np.ascontiguousarray((is_rob * wts).sum(axis=0), dtype=np.uint8)
            W = None
            if robin_weight is not None:
                W = _f64(robin_weight)
                if W.shape != (self.nd, self.nd, self.nf):
                    raise ValueError("robin_weight must have shape (nd, nd, Nf)")
            self._check(self.lib.pfv_mpsa_set_robin(self._h, _ptr(rbits, _up), _ptr(W, _dp)))
        if basis is not None:
            B = _f64(basis)
            if B.shape != (self.nd, self.nd, self.nf):
                raise ValueError("basis must have shape (nd, nd, Nf)")
            if not np.array_equal(B, np.tile(np.eye(self.nd)[:, :, None], (1, 1, self.nf))):
                self._check(self.lib.pfv_mpsa_set_basis(self._h, _ptr(B, _dp)))

    def mpsa_set_subface_eta(self, eta_subface):
        """``mpsa_eta`` per sub-face (face_nodes CSC order, sorted indices; mpsa.py:647-652); None removes it.
        After ``mpsa_set_params``."""
        es = None if eta_subface is None else _f64(eta_subface)
        if es is not None and es.shape != (self.nsf,):
            raise ValueError("size of eta must either be 1 or number of subfaces")
        self._check(self.lib.pfv_mpsa_set_subface_eta(self._h, _ptr(es, _dp)))

    def mpsa_set_reconstruction_eta(self, hf_eta):
        """``reconstruction_eta`` (mpsa.py:185, 757-761): where the displacement traces are reconstructed; None = at
        the continuity points; an array = one value per sub-face (sorted CSC order of face_nodes), used as given also
        on the boundary.  After ``mpsa_set_params``."""
        if hf_eta is not None and np.ndim(hf_eta) > 0 and np.size(hf_eta) != 1:
            e = _f64(np.asarray(hf_eta, dtype=float).ravel())
            if e.size != self.nsf:
                raise ValueError("reconstruction_eta per sub-face must have Nsf entries")
            self._check(self.lib.pfv_mpsa_set_reconstruction_eta_subface(self._h, _ptr(e, _dp)))
            return
        if hf_eta is not None:
            hf_eta = float(np.asarray(hf_eta).ravel()[0])
        self._check(self.lib.pfv_mpsa_set_reconstruction_eta(self._h, 0 if hf_eta is None else 1,
                                                             0.0 if hf_eta is None else float(hf_eta)))

    def mpsa_set_subface_bc(self, is_dir_sub, is_neu_sub, is_rob_sub=None, robin_weight_sub=None, basis_sub=None):
        """Conditions per sub-face (include/porefv.h: pfv_mpsa_set_subface_bc): boolean (nd, Nsf) arrays in the
        order of the sorted face_nodes CSC arrays, optional Robin weights (nd, nd, Nsf) and basis (nd, nd, Nsf:
        pfv_mpsa_set_subface_basis).  After mpsa_set_params."""
        is_dir, is_neu = np.asarray(is_dir_sub, bool), np.asarray(is_neu_sub, bool)
        if is_dir.shape != (self.nd, self.nsf) or is_neu.shape != (self.nd, self.nsf):
            raise ValueError("is_dir / is_neu per sub-face must have shape (nd, Nsf)")
        wts = (1 << np.arange(self.nd))[:, None]
        dbits = np.ascontiguousarray((is_dir * wts).sum(axis=0), dtype=np.uint8)
        nbits = np.ascontiguousarray((is_neu * wts).sum(axis=0), dtype=np.uint8)
        rbits = W = None
        if is_rob_sub is not None and np.any(is_rob_sub):
            is_rob = np.asarray(is_rob_sub, bool)
            if is_rob.shape != (self.nd, self.nsf):
                raise ValueError("is_rob per sub-face must have shape (nd, Nsf)")
            rbits = np.ascontiguousarray((is_rob * wts).sum(axis=0), dtype=np.uint8)
            if robin_weight_sub is not None:
                W = _f64(robin_weight_sub)
                if W.shape != (self.nd, self.nd, self.nsf):
                    raise ValueError("robin_weight per sub-face must have shape (nd, nd, Nsf)")
        self._check(self.lib.pfv_mpsa_set_subface_bc(self._h, _ptr(dbits, _up), _ptr(nbits, _up), _ptr(rbits, _up),
                                                     _ptr(W, _dp)))
        if basis_sub is not None:
            B = _f64(basis_sub)
            if B.shape != (self.nd, self.nd, self.nsf):
                raise ValueError("basis per sub-face must have shape (nd, nd, Nsf)")
            if not np.array_equal(B, np.broadcast_to(np.eye(self.nd)[:, :, None], B.shape)):
                self._check(self.lib.pfv_mpsa_set_subface_basis(self._h, _ptr(B, _dp)))

    # ---- Biot coupling terms ------------------------------------------------------------
    def biot_set_alphas(self, alphas):
        """alphas: sequence of (3, 3, Nc) coupling tensors (empty: switch the coupling terms off)."""
        arr = np.ascontiguousarray(np.stack([_f64(a) for a in alphas]) if len(alphas) else np.zeros((0, 3, 3, self.nc)))
        if arr.shape[1:] != (3, 3, self.nc):
            raise ValueError(f"coupling tensors must have shape (3, 3, {self.nc})")
        self._check(self.lib.pfv_biot_set_alphas(self._h, arr.shape[0], _ptr(arr, _dp) if arr.size else None))

    def biot_discretize(self, rebuild_topology=False):
        self._check(self.lib.pfv_biot_discretize(self._h, DISCR_REBUILD_TOPOLOGY if rebuild_topology else 0))
        self._discretized_m = True
        self._discretized_b = True

    def biot_discretize_faces(self, faces, cells, keep_other_rows: bool):
        """Partial discretization / update of MPSA + coupling terms (include/porefv.h)."""
        fa = np.ascontiguousarray(faces, dtype=np.int32)
        ce = np.ascontiguousarray(cells, dtype=np.int32)
        self._check(self.lib.pfv_biot_discretize_faces(self._h, 0, fa.size, _ptr(fa, _ip), ce.size, _ptr(ce, _ip),
                                                       1 if keep_other_rows else 0))
        self._discretized_m = self._discretized_m and bool(keep_other_rows)
        self._discretized_b = self._discretized_b and bool(keep_other_rows)

    @property
    def has_biot_discretization(self) -> bool:
        return self._discretized_b

    def biot_matrix(self, term: int, key: int):
        import scipy.sparse as sps

        r, c, z = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.pfv_biot_matrix_info(self._h, term, C.byref(r), C.byref(c), C.byref(z)))
        indptr = np.empty(r.value + 1, dtype=np.int32)
        indices = np.empty(z.value, dtype=np.int32)
        data = np.empty(z.value, dtype=np.float64)
        self._check(self.lib.pfv_biot_get_matrix(self._h, term, key, _ptr(indptr, _ip), _ptr(indices, _ip),
                                                 _ptr(data, _dp)))
        return sps.csr_matrix((data, indices, indptr), shape=(r.value, c.value))

    def mpsa_discretize(self, rebuild_topology=False):
        self._check(self.lib.pfv_mpsa_discretize(self._h, DISCR_REBUILD_TOPOLOGY if rebuild_topology else 0))
        self._discretized_m = True

    def mpsa_discretize_faces(self, faces, keep_other_rows: bool):
        fa = np.ascontiguousarray(faces, dtype=np.int32)
        self._check(self.lib.pfv_mpsa_discretize_faces(self._h, 0, fa.size, _ptr(fa, _ip),
                                                       1 if keep_other_rows else 0))
        self._discretized_m = self._discretized_m and bool(keep_other_rows)

    @property
    def has_mpsa_discretization(self) -> bool:
        return self._discretized_m

    def mpsa_assemble(self, bc_values, source=None):
        bcv = _f64(bc_values)
        if bcv.shape != (self.nf * self.nd,):
            raise ValueError("bc_values must have nd * Nf entries")
        src = None if source is None else _f64(source)
        self._check(self.lib.pfv_mpsa_assemble(self._h, _ptr(bcv, _dp), _ptr(src, _dp)))

    def free_device_bytes(self):
        """Free HBM of the handle's device in bytes (None from the host-emulation build)."""
        f, t = C.c_int64(), C.c_int64()
        self._check(self.lib.pfv_device_memory(self._h, C.byref(f), C.byref(t)))
        return None if f.value < 0 else int(f.value)

    def active_size(self) -> int:
        """Unknowns of the system assembled last (what solve / rhs read and write); 0 if none."""
        n = C.c_int64()
        self._check(self.lib.pfv_active_size(self._h, C.byref(n)))
        return int(n.value)

    def _active_n(self, n=None) -> int:
        """The library's own size of the active system; a caller-supplied n must agree with it (the
        output buffers are sized from this, never from caller input)."""
        m = self.active_size()
        if m <= 0:
            raise RuntimeError("no assembled system on this handle (assemble / set_system first)")
        if n is not None and int(n) != m:
            raise ValueError(f"the assembled system has {m} unknowns, not {int(n)}")
        return m

    def active_rhs(self, n=None):
        b = pinned_pool(self.lib).empty(self._active_n(n), np.float64)
        self._check(self.lib.pfv_get_rhs(self._h, _ptr(b, _dp)))
        return b

    # ---- hot path -----------------------------------------------------------------
    def discretize(self, rebuild_topology=False, skip_vector_source=False):
        flags = (DISCR_REBUILD_TOPOLOGY if rebuild_topology else 0) | \
                (DISCR_SKIP_VECTOR_SOURCE if skip_vector_source else 0)
        self._before_overwrite()
        self._check(self.lib.pfv_mpfa_discretize(self._h, flags))
        self._discretized = True

    def tpfa_discretize(self, vector_source_dim: int):
        self._before_overwrite()
        self._check(self.lib.pfv_tpfa_discretize(self._h, int(vector_source_dim)))
        self._discretized = False
        self._vs_dim = int(vector_source_dim)

    @property
    def has_discretization(self) -> bool:
        return self._discretized

    def discretize_faces(self, faces, keep_other_rows: bool, skip_vector_source=False):
        """Recompute the rows of ``faces`` only (partial discretization / update)."""
        fa = np.ascontiguousarray(faces, dtype=np.int32)
        flags = DISCR_SKIP_VECTOR_SOURCE if skip_vector_source else 0
        self._before_overwrite()
        self._check(self.lib.pfv_mpfa_discretize_faces(self._h, flags, fa.size, _ptr(fa, _ip),
                                                       1 if keep_other_rows else 0))
        self._discretized = self._discretized and bool(keep_other_rows)

    def matrix_info(self, which: int):
        r, c, z = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.pfv_matrix_info(self._h, which, C.byref(r), C.byref(c), C.byref(z)))
        return r.value, c.value, z.value

    def matrix_rows(self, which: int, rows) -> "sps.csr_matrix":
        """Rows ``rows`` of a result matrix as a (len(rows) x ncols) scipy csr: gathered on the device, only
        those entries cross PCIe."""
        import scipy.sparse as sps

        _, ncols, _ = self.matrix_info(which)
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        indptr = np.empty(rows.size + 1, dtype=np.int32)
        self._check(self.lib.pfv_get_matrix_rows(self._h, which, rows.size, _ptr(rows, _ip), _ptr(indptr, _ip), None, None))
        indices = np.empty(max(int(indptr[-1]), 1), dtype=np.int32)
        data = np.empty(max(int(indptr[-1]), 1), dtype=np.float64)
        self._check(self.lib.pfv_get_matrix_rows(self._h, which, rows.size, _ptr(rows, _ip), _ptr(indptr, _ip),
                                                 _ptr(indices, _ip), _ptr(data, _dp)))
        nnz = int(indptr[-1])
        return sps.csr_matrix((data[:nnz], indices[:nnz], indptr), shape=(rows.size, ncols))

    def matrix(self, which: int, rows=None):
        """Copy a result matrix to the host as scipy csr (int32 sorted indices, FP64).
        ``rows``: keep only these rows (all others come back empty, the shape is unchanged)."""
        import scipy.sparse as sps

        nrows, ncols, nnz = self.matrix_info(which)
        if which in (MAT_VECTOR_SOURCE, MAT_BOUND_PRESSURE_VECTOR_SOURCE) and (
                nnz >= 2 ** 31 or os.environ.get("PFV_VS_IMPLICIT", "0") not in ("", "0")):
            # beyond 2^31 entries (nd x the flux pattern: ~6.3 M tetrahedra) the handle keeps no int32 CSR arrays of the two
            # vector-source matrices (csrc/topology.inc: vs_implicit): fetched in row chunks, joined with int64 row pointers
            # (scipy widens its index arrays by itself)
            want = None if rows is None else np.unique(np.asarray(rows, dtype=np.int64))
            step = max(1, int(2 ** 29 // max(1, nnz // max(nrows, 1))))
            parts, ptr = [], [np.zeros(1, dtype=np.int64)]
            for r0 in range(0, nrows, step):
                r1 = min(nrows, r0 + step)
                rr = np.arange(r0, r1) if want is None else want[(want >= r0) & (want < r1)]
                lens = np.zeros(r1 - r0, dtype=np.int64)
                if rr.size:
                    M = self.matrix_rows(which, rr)
                    parts.append((M.indices.astype(np.int64 if ncols >= 2 ** 31 else np.int32), M.data))
                    lens[rr - r0] = np.diff(M.indptr)
                ptr.append(ptr[-1][-1] + np.cumsum(lens))
            indptr = np.concatenate(ptr)
            indices = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int32)
            data = np.concatenate([p[1] for p in parts]) if parts else np.zeros(0)
            return sps.csr_matrix((data, indices, indptr), shape=(nrows, ncols))
        pool = pinned_pool(self.lib)  # page-locked blocks, recycled from the matrices that were collected
        indptr = pool.empty(nrows + 1, np.int32)
        indices = pool.empty(nnz, np.int32)
        data = pool.empty(nnz, np.float64)
        self._check(self.lib.pfv_get_matrix(self._h, which, _ptr(indptr, _ip), _ptr(indices, _ip),
                                            _ptr(data, _dp)))
        if rows is not None:
            keep = np.zeros(nrows, dtype=bool)
            keep[np.asarray(rows, dtype=np.int64)] = True
            lens = np.where(keep, np.diff(indptr), 0).astype(np.int32)
            sel = np.repeat(keep, np.diff(indptr))
            indices, data = indices[sel], data[sel]
            indptr = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
        return sps.csr_matrix((data, indices, indptr), shape=(nrows, ncols))

    def assemble(self, bc_values, vector_source=None, source=None):
        bcv = _f64(bc_values)
        if bcv.shape != (self.nf,):
            raise ValueError("bc_values must have one entry per face")
        vs = None if vector_source is None else _f64(vector_source)
        if vs is not None and vs.shape != (self.matrix_info(MAT_VECTOR_SOURCE)[1],):
            raise ValueError("vector_source must have one entry per column of the vector_source matrix")
        src = None if source is None else _f64(source)
        if src is not None and src.shape != (self.nc,):
            raise ValueError("source must have one entry per cell")
        self._check(self.lib.pfv_mpfa_assemble(self._h, _ptr(bcv, _dp), _ptr(vs, _dp), _ptr(src, _dp)))

    def set_periodic(self, native_cell, shift):
        """Displaced sides of merged periodic faces (include/porefv.h: pfv_set_periodic)."""
        nat = np.ascontiguousarray(native_cell, dtype=np.int32)
        sh = np.ascontiguousarray(shift, dtype=np.float64)
        if nat.shape != (self.nf,) or sh.shape != (3, self.nf):
            raise ValueError("native_cell must be (Nf,), shift (3, Nf)")
        self._check(self.lib.pfv_set_periodic(self._h, nat.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(sh, _dp)))

    # ---- device-resident vectors (addresses of device buffers, e.g. torch.Tensor.data_ptr()) ----
    def _dev(self, on: bool):
        self._check(self.lib.pfv_set_vectors_on_device(self._h, 1 if on else 0))

    def assemble_device(self, bc_ptr: int, vs_ptr: int = 0, src_ptr: int = 0):
        """``assemble`` with device buffers: Nf bc values, optional vector source, optional Nc sources."""
        self._dev(True)
        try:
            self._check(self.lib.pfv_mpfa_assemble(self._h, C.cast(bc_ptr, _dp), C.cast(vs_ptr or None, _dp),
                                                   C.cast(src_ptr or None, _dp)))
        finally:
            self._dev(False)

    def mpsa_assemble_device(self, bc_ptr: int, src_ptr: int = 0):
        self._dev(True)
        try:
            self._check(self.lib.pfv_mpsa_assemble(self._h, C.cast(bc_ptr, _dp), C.cast(src_ptr or None, _dp)))
        finally:
            self._dev(False)

    def solve_device(self, x_ptr: int, method="bicgstab", rtol=1e-12, maxit=10000, x0_ptr: int = 0,
                     raise_on_fail=True, restart=0, precond="jacobi"):
        """``solve`` writing the solution into the device buffer at ``x_ptr`` (n doubles); returns info."""
        code = {"cg": SOLVE_CG, "bicgstab": SOLVE_BICGSTAB, "gmres": SOLVE_GMRES}[method]
        self._check(self.lib.pfv_set_preconditioner(self._h, {"jacobi": 0, "amg": 1, "block": 2}[precond]))
        info = SolveInfo()
        self._dev(True)
        try:
            st = self.lib.pfv_solve(self._h, code, float(rtol), int(maxit), int(restart), C.cast(x0_ptr or None, _dp),
                                    C.cast(x_ptr, _dp), C.byref(info))
        finally:
            self._dev(False)
        out = {"iterations": info.iterations, "converged": bool(info.converged),
               "rel_residual": info.rel_residual, "solve_ms": info.solve_ms}
        if st != 0 and (raise_on_fail or st != 6):
            self._check(st)
        return out

    def rhs(self):
        return self.active_rhs()

    def spmv(self, which: int, x):
        nrows, ncols, _ = self.matrix_info(which)
        x = _f64(x)
        if x.shape != (ncols,):
            raise ValueError("dimension mismatch")
        y = np.empty(nrows, dtype=np.float64)
        self._check(self.lib.pfv_spmv(self._h, which, _ptr(x, _dp), _ptr(y, _dp)))
        return y

    def solve(self, method="bicgstab", rtol=1e-12, maxit=10000, x0=None, raise_on_fail=True, n=None,
              restart=0, precond="jacobi"):
        """Solve the system assembled last (flow: n = Nc; mechanics: pass n = nd * Nc).
        ``restart``: GMRES cycle length (0 = 30); ``precond``: "jacobi" or "amg"."""
        code = {"cg": SOLVE_CG, "bicgstab": SOLVE_BICGSTAB, "gmres": SOLVE_GMRES}[method]
        self._check(self.lib.pfv_set_preconditioner(self._h, {"jacobi": 0, "amg": 1, "block": 2}[precond]))
        x = pinned_pool(self.lib).empty(self._active_n(n), np.float64)
        x0a = None if x0 is None else _f64(x0)
        if x0a is not None and x0a.shape != x.shape:
            raise ValueError("x0 has the wrong length")
        info = SolveInfo()
        st = self.lib.pfv_solve(self._h, code, float(rtol), int(maxit), int(restart), _ptr(x0a, _dp), _ptr(x, _dp),
                                C.byref(info))
        out = {"iterations": info.iterations, "converged": bool(info.converged),
               "rel_residual": info.rel_residual, "solve_ms": info.solve_ms}
        if st != 0 and (raise_on_fail or st != 6):
            self._check(st)
        return x, out

    def set_system(self, A, b):
        """Hand an assembled scipy CSR system to the device solver (no grid needed)."""
        import scipy.sparse as sps

        A = sps.csr_matrix(A)
        n = A.shape[0]
        if A.shape[0] != A.shape[1] or np.asarray(b).shape != (n,):
            raise ValueError("square matrix and matching right-hand side expected")
        if A.nnz >= 2 ** 31:
            raise ValueError("more than 2^31 matrix entries")
        ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
        ix = np.ascontiguousarray(A.indices, dtype=np.int32)
        dv = _f64(A.data)
        bb = _f64(b)
        self._check(self.lib.pfv_set_system(self._h, n, _ptr(ip, _ip), _ptr(ix, _ip), _ptr(dv, _dp), _ptr(bb, _dp)))
        self._user_n = n

    def set_block_preconditioner(self, block_ptr, gauss_seidel: bool = True):
        """Contiguous blocks [block_ptr[k], block_ptr[k+1]) of the system of ``set_system``: the following solves with
        ``precond="block"`` use the block lower-triangular preconditioner (pfv_set_block_preconditioner)."""
        bp = np.ascontiguousarray(block_ptr, dtype=np.int64)
        self._check(self.lib.pfv_set_block_preconditioner(self._h, bp.size - 1, bp.ctypes.data_as(C.POINTER(C.c_int64)),
                                                          1 if gauss_seidel else 0))

    def stats(self) -> dict:
        s = Stats()
        self._check(self.lib.pfv_get_stats_n(self._h, C.byref(s), C.sizeof(s)))
        return s.as_dict()

    def sync(self):
        self._check(self.lib.pfv_sync(self._h))

    # ---- device-pointer entry points (vectors owned by the caller, e.g. torch tensors) ----
    def spmv_device_rows(self, which: int, nrows: int, x_ptr: int, y_ptr: int):
        self._check(self.lib.pfv_spmv_device_rows(self._h, which, int(nrows), C.c_void_p(x_ptr), C.c_void_p(y_ptr)))

    def copy_device_vector(self, which: int, dst_ptr: int, count: int):
        self._check(self.lib.pfv_copy_device_vector(self._h, which, C.c_void_p(dst_ptr), int(count)))

    def amg_setup(self, n_own: int = 0):
        """AMG hierarchy of the leading n_own x n_own block of the assembled system (0 = all of it)."""
        self._check(self.lib.pfv_amg_setup(self._h, int(n_own)))

    def amg_setup_sharded(self, n_own: int, hooks: "ShardHooks", rank: int, world: int, peers, send_lists, recv_lists):
        """Coupled hierarchy of a sharded solve (pfv_amg_setup_sharded): ``send_lists[i]`` are the owned CELLS peer
        ``peers[i]`` needs (in the order it expects them), ``recv_lists[i]`` the local cells (>= n_own / bs) its values
        land in.  A collective: every rank of the group calls it.  ``hooks`` must carry sendrecv and allgather and is
        kept alive by this handle until the next setup."""
        peers = np.ascontiguousarray(peers, dtype=np.int32)
        sp = np.zeros(len(peers) + 1, dtype=np.int64)
        rp = np.zeros(len(peers) + 1, dtype=np.int64)
        for i in range(len(peers)):
            sp[i + 1] = sp[i] + len(send_lists[i])
            rp[i + 1] = rp[i] + len(recv_lists[i])
        cat = lambda ls: (np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.int32) for a in ls]), dtype=np.int32)
                          if len(ls) else np.zeros(0, dtype=np.int32))
        si, rpos = cat(send_lists), cat(recv_lists)
        self._coupled_hooks = hooks  # (the C side copies the struct; the callbacks behind it must outlive the hierarchy)
        self._check(self.lib.pfv_amg_setup_sharded(
            self._h, int(n_own), C.byref(hooks), int(rank), int(world), int(len(peers)),
            peers.ctypes.data_as(_ip), sp.ctypes.data_as(_lp), si.ctypes.data_as(_ip), rp.ctypes.data_as(_lp),
            rpos.ctypes.data_as(_ip)))

    def amg_apply_device(self, r_ptr: int, z_ptr: int):
        """z = V-cycle(r) on device vectors (length of the block given to amg_setup)."""
        self._check(self.lib.pfv_amg_apply_device(self._h, C.c_void_p(r_ptr), C.c_void_p(z_ptr)))

    def solve_sharded(self, n_own: int, exchange_halo, allreduce_sum, work_ptr: int, x_ptr: int,
                      method="bicgstab", rtol=1e-10, maxit=20000, precond="jacobi", raise_on_fail=False):
        """The fused Krylov loop on the leading ``n_own`` rows of the assembled system (pfv_solve_sharded).
        ``exchange_halo(d_x_ptr)`` fills the halo entries of the SpMV input at that address,
        ``allreduce_sum(d_vals_ptr, count)`` sums ``count`` doubles over the ranks in place; both only
        enqueue work on the handle's stream.  ``work_ptr``: 2 n_local + 8 doubles of device memory (the
        addresses the hooks see point into it), ``x_ptr``: n_own doubles for the solution."""
        code = {"cg": SOLVE_CG, "bicgstab": SOLVE_BICGSTAB}[method]
        self._check(self.lib.pfv_set_preconditioner(self._h, {"jacobi": 0, "amg": 1, "block": 2}[precond]))
        failure = []

        def _halo(_user, d_x, _stream):
            try:
                exchange_halo(int(d_x))
                return 0
            except BaseException as e:  # must not propagate through the C frames
                failure.append(e)
                return 1

        def _red(_user, d_vals, count, _stream):
            try:
                allreduce_sum(int(d_vals), int(count))
                return 0
            except BaseException as e:
                failure.append(e)
                return 1

        if isinstance(exchange_halo, RcclComm):
            # native transport: the hooks are C functions of the library (rccl_hooks.inc), nothing of the
            # iteration runs in Python
            hooks = ShardHooks()
            self._check(self.lib.pfv_rccl_hooks(exchange_halo._c, C.byref(hooks)))
        else:
            hooks = ShardHooks(HALO_FN(_halo), ALLREDUCE_FN(_red), None)
        info = SolveInfo()
        st = self.lib.pfv_solve_sharded(self._h, code, float(rtol), int(maxit), int(n_own), C.byref(hooks),
                                        C.c_void_p(work_ptr), C.c_void_p(x_ptr), C.byref(info))
        if failure:
            raise failure[0]
        if isinstance(exchange_halo, RcclComm) and st not in (0, 6):
            msg = exchange_halo.last_error()
            if msg:
                raise PorefvError(st, "RCCL transport: " + msg)
        out = {"iterations": info.iterations, "converged": bool(info.converged),
               "rel_residual": info.rel_residual, "solve_ms": info.solve_ms}
        if st != 0 and (raise_on_fail or st != 6):
            self._check(st)
        return out

    def ad_flux_system(self, p, dk_dp=None, bc_values=None, vector_source=None, source=None, flux_jacobian=False):
        """Residual and Jacobian of the flow equation with K = K(p) (pfv_mpfa_ad_flux_system), left on the device
        as the active system (``solve`` returns the Newton increment; ``matrix(MAT_SYSTEM)`` / ``rhs()`` copy J
        and -r out).  Returns the face fluxes q; with ``flux_jacobian`` dq/dp is ``matrix(MAT_FLUX_JACOBIAN)``."""
        pa_ = _f64(p)
        if pa_.shape != (self.nc,):
            raise ValueError("p must have one entry per cell")
        dk = None if dk_dp is None else _f64(dk_dp)
        if dk is not None and dk.shape != (3, 3, self.nc):
            raise ValueError("dk_dp must have shape (3, 3, num_cells)")
        bc = None if bc_values is None else _f64(bc_values)
        vs = None if vector_source is None else _f64(vector_source)
        src = None if source is None else _f64(source)
        q = np.empty(self.nf, dtype=np.float64)
        self._check(self.lib.pfv_mpfa_ad_flux_system(self._h, _ptr(pa_, _dp), _ptr(dk, _dp), _ptr(bc, _dp), _ptr(vs, _dp),
                                                     _ptr(src, _dp), _ptr(q, _dp), 1 if flux_jacobian else 0))
        return q

    def tpfa_transmissibility_ad(self, perm):
        """Two-point face transmissibilities and their derivatives with respect to the permeability entries
        of the neighbouring cells (include/porefv.h: pfv_tpfa_transmissibility_ad).  Returns
        (t_face (Nf,), dt_dk (nnz(cell_faces), 9)) in the order of the cell_faces entries."""
        K = _f64(perm)
        if K.shape != (3, 3, self.nc):
            raise ValueError("perm must have shape (3, 3, num_cells)")
        t = np.empty(self.nf, dtype=np.float64)
        jac = np.empty((self.ncf, 9), dtype=np.float64)
        self._check(self.lib.pfv_tpfa_transmissibility_ad(self._h, _ptr(K, _dp), _ptr(t, _dp), _ptr(jac, _dp)))
        return t, jac

    def set_stream(self, stream_ptr: int | None):
        """Run on the caller's HIP stream (0 / None = the legacy default stream, torch's default)."""
        self._check(self.lib.pfv_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def reset_stream(self):
        self._check(self.lib.pfv_reset_stream(self._h))

    def debug_array(self, which: int, n: int) -> np.ndarray:
        """Leading n entries of an internal per-node array (0: response tables, 1: boundary columns), for tests."""
        out = np.empty(n, dtype=np.float64)
        self._check(self.lib.pfv_debug_copy(self._h, int(which), _ptr(out, _dp), n))
        return out

    def time_kernel(self, kernel: int, reps: int = 10) -> float:
        """Average ms per launch (HIP events on the handle's stream); 0 SpMV(A), 1 node, 2 face."""
        ms = C.c_double()
        self._check(self.lib.pfv_time_kernel(self._h, int(kernel), int(reps), C.byref(ms)))
        return ms.value
