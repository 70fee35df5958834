//! Raw bindings of `libczk_hip.so` -- GENERATED from include/czk.h by tools/gen_rust_sys.py; do not edit.
//! One `extern "C"` declaration per C declaration; constants mirror the C enums.  Safe wrappers live in the `czk` crate.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_long, c_uint, c_void};

#[repr(C)]
pub struct czk_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czk_bases {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czk_lanes {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czk_net {
    _private: [u8; 0],
}
#[repr(C)]
pub struct czk_r1cs_matrix {
    _private: [u8; 0],
}

pub const CZK_OK: c_int = 0; // czk_status
pub const CZK_ERR_SIZE: c_int = 1; // czk_status
pub const CZK_ERR_HIP: c_int = 2; // czk_status
pub const CZK_ERR_ARG: c_int = 3; // czk_status
pub const CZK_ERR_NOMEM: c_int = 4; // czk_status
pub const CZK_ERR_NET: c_int = 5; // czk_status
pub const CZK_ERR_CHECK: c_int = 6; // czk_status
pub const CZK_MEM_HOST: c_int = 0; // czk_mem
pub const CZK_MEM_DEVICE: c_int = 1; // czk_mem
pub const CZK_MEM_STABLE: c_int = 16; // czk_mem
pub const CZK_MEM_NO_TABLES: c_int = 32; // czk_mem
pub const CZK_MEM_ANY_POINTS: c_int = 64; // czk_mem
pub const CZK_MEM_CHECK_SUBGROUP: c_int = 128; // czk_mem
pub const CZK_MEM_SCALAR_HOST: c_int = 256; // czk_mem
pub const CZK_MEM_SAME_SCALARS: c_int = 512; // czk_mem
pub const CZK_FFT: c_int = 0; // czk_ntt_kind
pub const CZK_IFFT: c_int = 1; // czk_ntt_kind
pub const CZK_COSET_FFT: c_int = 2; // czk_ntt_kind
pub const CZK_COSET_IFFT: c_int = 3; // czk_ntt_kind
pub const CZK_SCALAR_CANONICAL: c_int = 0; // czk_scalar_form
pub const CZK_SCALAR_MONTGOMERY: c_int = 1; // czk_scalar_form
pub const CZK_G1: c_int = 1; // czk_group
pub const CZK_G2: c_int = 2; // czk_group
pub const CZK_OP_ADD: c_int = 0; // czk_binop
pub const CZK_OP_SUB: c_int = 1; // czk_binop
pub const CZK_OP_MUL: c_int = 2; // czk_binop
pub const CZK_NET_RCCL: c_int = 1; // czk_net_transport
pub const CZK_NET_SHM: c_int = 2; // czk_net_transport
pub const CZK_NET_IPC: c_int = 3; // czk_net_transport
pub const CZK_NET_UNIQUE_ID_BYTES: c_int = 128; // #define
pub const CZK_OPEN_COMMIT: c_int = 1; // #define

#[link(name = "czk_hip")]
extern "C" {
    pub fn czk_ctx_create(out: *mut *mut czk_ctx, device: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn czk_ctx_destroy(ctx: *mut czk_ctx);
    pub fn czk_ctx_sync(ctx: *mut czk_ctx) -> c_int;
    pub fn czk_ctx_mark(ctx: *mut czk_ctx, out_mark: *mut u64) -> c_int;
    pub fn czk_ctx_wait_mark(ctx: *mut czk_ctx, mark: u64) -> c_int;
    pub fn czk_ctx_stream(ctx: *const czk_ctx) -> *mut c_void;
    pub fn czk_ctx_reserve(ctx: *mut czk_ctx, ntt_log_d: c_uint, ntt_lanes: usize, bases: *const czk_bases, n_scalars: usize, msm_lanes: usize) -> c_int;
    pub fn czk_last_error(ctx: *const czk_ctx) -> *const c_char;
    pub fn czk_version() -> *const c_char;
    pub fn czk_ctx_set_option(ctx: *mut czk_ctx, name: *const c_char, value: c_long) -> c_int;
    pub fn czk_build_is_lab() -> c_int;
    pub fn czk_lanes_alloc(ctx: *mut czk_ctx, lanes: usize, len: usize, out: *mut *mut czk_lanes) -> c_int;
    pub fn czk_lanes_free(l: *mut czk_lanes);
    pub fn czk_lanes_count(l: *const czk_lanes) -> usize;
    pub fn czk_lanes_len(l: *const czk_lanes) -> usize;
    pub fn czk_lanes_data(l: *const czk_lanes, lane: usize, elem: usize) -> *mut u64;
    pub fn czk_lanes_upload(ctx: *mut czk_ctx, dst: *mut czk_lanes, lane: usize, elem: usize, host: *const u64, n: usize) -> c_int;
    pub fn czk_lanes_download(ctx: *mut czk_ctx, src: *const czk_lanes, lane: usize, elem: usize, host: *mut u64, n: usize) -> c_int;
    pub fn czk_lanes_download_deferred(ctx: *mut czk_ctx, src: *const czk_lanes, lane: usize, elem: usize, host: *mut u64, n: usize) -> c_int;
    pub fn czk_lanes_copy(ctx: *mut czk_ctx, dst: *mut czk_lanes, dst_lane: usize, dst_elem: usize, src: *const czk_lanes, src_lane: usize, src_elem: usize, n: usize) -> c_int;
    pub fn czk_lanes_zero(ctx: *mut czk_ctx, dst: *mut czk_lanes, lane: usize, elem: usize, n: usize) -> c_int;
    pub fn czk_fr_copy_3d(ctx: *mut czk_ctx, dst: *mut u64, dst_stride: *const usize, src: *const u64, src_stride: *const usize, n: *const usize) -> c_int;
    pub fn czk_ntt_fr(ctx: *mut czk_ctx, data: *mut u64, log_d: c_uint, lanes: usize, kind: c_int, in_len: usize, mem: c_int) -> c_int;
    pub fn czk_ntt_fr_to(ctx: *mut czk_ctx, src: *const u64, src_stride: usize, dst: *mut u64, log_d: c_uint, lanes: usize, kind: c_int, in_len: usize, mem: c_int) -> c_int;
    pub fn czk_domain_constants(ctx: *mut czk_ctx, log_d: c_uint, out24: *mut u64) -> c_int;
    pub fn czk_ntt_fr_mixed(ctx: *mut czk_ctx, data: *mut u64, size: usize, lanes: usize, kind: c_int, in_len: usize, mem: c_int) -> c_int;
    pub fn czk_mixed_domain_constants(ctx: *mut czk_ctx, size: usize, out24: *mut u64) -> c_int;
    pub fn czk_fr_vec_op(ctx: *mut czk_ctx, op: c_int, a: *const u64, b: *const u64, out: *mut u64, n: usize, mem: c_int) -> c_int;
    pub fn czk_fr_vec_scale(ctx: *mut czk_ctx, a: *const u64, k: *const u64, out: *mut u64, n: usize, mem: c_int) -> c_int;
    pub fn czk_fr_powers(ctx: *mut czk_ctx, g: *const u64, c: *const u64, n: usize, out: *mut u64, mem: c_int) -> c_int;
    pub fn czk_fr_beaver_combine(ctx: *mut czk_ctx, x: *const u64, y: *const u64, z: *const u64, sx: *const u64, oy: *const u64, add_open: c_int, out: *mut u64, n: usize, mem: c_int) -> c_int;
    pub fn czk_fr_spdz_open(ctx: *mut czk_ctx, shares: *const u64, parties: usize, n: usize, out_value: *mut u64, out_bad: *mut u64) -> c_int;
    pub fn czk_fr_lanes_sum(ctx: *mut czk_ctx, x: *const u64, k: usize, n: usize, out: *mut u64, out_nonzero: *mut u64) -> c_int;
    pub fn czk_fr_spdz_dx(ctx: *mut czk_ctx, value: *const u64, mac: *const u64, mac_share: *const u64, out: *mut u64, n: usize) -> c_int;
    pub fn czk_share_domain_constants(ctx: *mut czk_ctx, parties: usize, out12: *mut u64) -> c_int;
    pub fn czk_fr_gsz_open(ctx: *mut czk_ctx, shares: *const u64, parties: usize, n: usize, degrees: *const u32, degree: c_uint, out_value: *mut u64, out_bad: *mut u64) -> c_int;
    pub fn czk_net_unique_id(transport: c_int, out: *mut u8, cap: usize, len: *mut usize) -> c_int;
    pub fn czk_net_create(ctx: *mut czk_ctx, transport: c_int, rank: c_int, world: c_int, id: *const u8, id_len: usize, out: *mut *mut czk_net) -> c_int;
    pub fn czk_net_destroy(net: *mut czk_net);
    pub fn czk_net_rank(net: *const czk_net) -> c_int;
    pub fn czk_net_world(net: *const czk_net) -> c_int;
    pub fn czk_net_set_option(net: *mut czk_net, name: *const c_char, value: c_long) -> c_int;
    pub fn czk_net_last_error(net: *const czk_net) -> *const c_char;
    pub fn czk_net_stats(net: *const czk_net, out5: *mut u64) -> c_int;
    pub fn czk_net_stats_reset(net: *mut czk_net);
    pub fn czk_net_broadcast(net: *mut czk_net, send: *const c_void, bytes: usize, recv: *mut c_void, mem: c_int) -> c_int;
    pub fn czk_net_send_to_king(net: *mut czk_net, send: *const c_void, bytes: usize, recv: *mut c_void, mem: c_int) -> c_int;
    pub fn czk_net_recv_from_king(net: *mut czk_net, send: *const c_void, bytes: usize, recv: *mut c_void, mem: c_int) -> c_int;
    pub fn czk_net_barrier(net: *mut czk_net) -> c_int;
    pub fn czk_net_atomic_broadcast(net: *mut czk_net, x: *const u64, n: usize, recv: *mut u64, rand32: *const u8, mem: c_int) -> c_int;
    pub fn czk_spdz_batch_open(net: *mut czk_net, sh: *const u64, mac: *const u64, mac_share: *const u64, n: usize, out_value: *mut u64, flags: c_int, out_bad: *mut u64) -> c_int;
    pub fn czk_add_batch_open(net: *mut czk_net, val: *const u64, n: usize, out_value: *mut u64) -> c_int;
    pub fn czk_gsz_batch_open(net: *mut czk_net, val: *const u64, n: usize, degrees: *const u32, degree: c_uint, out_value: *mut u64, out_bad: *mut u64) -> c_int;
    pub fn czk_fr_send_to_king(net: *mut czk_net, x: *const u64, n: usize, gathered: *mut u64) -> c_int;
    pub fn czk_fr_recv_from_king(net: *mut czk_net, parts: *const u64, n: usize, out: *mut u64) -> c_int;
    pub fn czk_gsz_batch_king_compute(net: *mut czk_net, val: *const u64, n: usize, degrees: *const u32, degree: c_uint, out: *mut u64, out_bad: *mut u64) -> c_int;
    pub fn czk_fr_vec_serialize(ctx: *mut czk_ctx, a: *const u64, n: usize, mem: c_int, out: *mut u8) -> c_int;
    pub fn czk_fr_vec_deserialize(ctx: *mut czk_ctx, bytes: *const u8, len: usize, out: *mut u64, cap: usize, mem: c_int, n: *mut usize) -> c_int;
    pub fn czk_sha256(data: *const c_void, len: usize, out32: *mut u8);
    pub fn czk_r1cs_matrix_register(ctx: *mut czk_ctx, row_ptr: *const u64, col_idx: *const u32, coeff: *const u64, m: usize, nnz: usize, n_vars: usize, mem: c_int, out: *mut *mut czk_r1cs_matrix) -> c_int;
    pub fn czk_r1cs_matrix_release(a: *mut czk_r1cs_matrix);
    pub fn czk_r1cs_matvec(ctx: *mut czk_ctx, a: *const czk_r1cs_matrix, z: *const u64, z_stride: usize, lanes: usize, out: *mut u64, out_stride: usize, mem: c_int) -> c_int;
    pub fn czk_poly_div_linear(ctx: *mut czk_ctx, coeffs: *const u64, n: usize, lanes: usize, z: *const u64, quotient: *mut u64, remainder: *mut u64, mem: c_int) -> c_int;
    pub fn czk_poly_evaluate(ctx: *mut czk_ctx, coeffs: *const u64, n: usize, lanes: usize, z: *const u64, values: *mut u64, mem: c_int) -> c_int;
    pub fn czk_poly_evaluate_many(ctx: *mut czk_ctx, count: usize, coeffs: *const *const u64, n: *const usize, lanes: *const usize, z: *const u64, values: *const *mut u64) -> c_int;
    pub fn czk_fr_lincomb(ctx: *mut czk_ctx, count: usize, terms: *const *const u64, term_len: *const usize, term_lanes: *const usize, coeffs: *const u64, constant: *const u64, lanes: usize, lift_mask: u64, out: *mut u64, out_len: usize) -> c_int;
    pub fn czk_poly_div_vanishing(ctx: *mut czk_ctx, coeffs: *const u64, m: usize, lanes: usize, n: usize, quotient: *mut u64, remainder: *mut u64, mem: c_int) -> c_int;
    pub fn czk_fr_prefix_product(ctx: *mut czk_ctx, x: *const u64, n: usize, out: *mut u64, mem: c_int) -> c_int;
    pub fn czk_fr_batch_inverse(ctx: *mut czk_ctx, v: *const u64, n: usize, coeff: *const u64, out: *mut u64, mem: c_int) -> c_int;
    pub fn czk_fr_into_repr(ctx: *mut czk_ctx, a: *const u64, out: *mut u64, n: usize, mem: c_int) -> c_int;
    pub fn czk_fr_from_repr(ctx: *mut czk_ctx, a: *const u64, out: *mut u64, n: usize, mem: c_int) -> c_int;
    pub fn czk_bases_register(ctx: *mut czk_ctx, group: c_int, bases: *const u64, inf: *const u8, n: usize, mem: c_int, out: *mut *mut czk_bases) -> c_int;
    pub fn czk_bases_release(b: *mut czk_bases);
    pub fn czk_bases_len(b: *const czk_bases) -> usize;
    pub fn czk_bases_layout(b: *const czk_bases, c: *mut c_uint, windows: *mut c_uint) -> c_int;
    pub fn czk_bases_layout_for(b: *const czk_bases, n_scalars: usize, c: *mut c_uint, windows: *mut c_uint) -> c_int;
    pub fn czk_bases_arith(b: *const czk_bases) -> c_int;
    pub fn czk_bases_prepare(ctx: *mut czk_ctx, b: *const czk_bases, n_scalars: usize) -> c_int;
    pub fn czk_msm(ctx: *mut czk_ctx, bases: *const czk_bases, scalars: *const u64, n_scalars: usize, lanes: usize, scalar_form: c_int, mem: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_msm_async(ctx: *mut czk_ctx, bases: *const czk_bases, scalars: *const u64, n_scalars: usize, lanes: usize, scalar_form: c_int, mem: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_msm_g1(ctx: *mut czk_ctx, bases_xy: *const u64, inf: *const u8, scalars: *const u64, n: usize, lanes: usize, scalar_form: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_msm_g2(ctx: *mut czk_ctx, bases_xy: *const u64, inf: *const u8, scalars: *const u64, n: usize, lanes: usize, scalar_form: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_bases_check_subgroup(ctx: *mut czk_ctx, bases: *const czk_bases, out_bad: *mut usize) -> c_int;
    pub fn czk_jac_to_affine(ctx: *mut czk_ctx, group: c_int, jac: *const u64, n: usize, out_aff: *mut u64, out_inf: *mut u8) -> c_int;
    pub fn czk_jac_add(ctx: *mut czk_ctx, group: c_int, a_jac: *const u64, b_jac: *const u64, out_jac: *mut u64) -> c_int;
    pub fn czk_jac_add_mixed(ctx: *mut czk_ctx, group: c_int, a_jac: *const u64, b_aff: *const u64, b_inf: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_jac_scalar_mul(ctx: *mut czk_ctx, group: c_int, a_jac: *const u64, k: *const u64, scalar_form: c_int, out_jac: *mut u64) -> c_int;
    pub fn czk_jac_neg(ctx: *mut czk_ctx, group: c_int, a_jac: *const u64, out_jac: *mut u64) -> c_int;
    pub fn czk_fixed_base_points(ctx: *mut czk_ctx, group: c_int, k: *const u64, n: usize, out: *mut u64, mem: c_int) -> c_int;
    pub fn czk_witness_map_pre(ctx: *mut czk_ctx, a: *mut u64, a_len: usize, b: *mut u64, b_len: usize, log_d: c_uint, lanes: usize) -> c_int;
    pub fn czk_witness_map_post(ctx: *mut czk_ctx, ab: *mut u64, c: *mut u64, c_len: usize, log_d: c_uint, lanes: usize) -> c_int;
    pub fn czk_profile_enable(ctx: *mut czk_ctx, on: c_int) -> c_int;
    pub fn czk_profile_reset(ctx: *mut czk_ctx) -> c_int;
    pub fn czk_profile_read(ctx: *mut czk_ctx, kernel: *const c_char, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn czk_profile_intervals(ctx: *mut czk_ctx, kernel: *const c_char, start_ms: *mut f64, stop_ms: *mut f64, cap: usize, n: *mut usize) -> c_int;
    pub fn czk_profile_base_offset(a: *mut czk_ctx, b: *mut czk_ctx, ms: *mut f64) -> c_int;
}
