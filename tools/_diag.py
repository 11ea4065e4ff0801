"""tools/ only: route the package's ops through the DIAGNOSTIC library (tests/_diag/libdclnet_hip_diag.so, `make -C
dcl-net_amd/csrc diag`; DCL_HIP_LIB=<path> picks another diagnostic build, e.g. the stamps library) -- the product library has no
dcl_debug_* hooks.  The environment switches below are read here, in the tool, never by the product."""
import os


def use_diag(dcl):
    L = dcl._native.diagnostic_library(os.environ.get("DCL_HIP_LIB")).__enter__()
    for env, fn in (("DCL_CONV_VARIANT", "dcl_debug_force_valu_conv"), ("DCL_CONV_XCD", "dcl_debug_conv_xcd_remap"),
                    ("DCL_CONV_SPLIT", "dcl_debug_conv_split"), ("DCL_CONV_FEW", "dcl_debug_conv_few_chunks"), ("DCL_CONV_FEW_TILES", "dcl_debug_conv_few_tiles"), ("DCL_CONV_WLDS", "dcl_debug_conv_wlds"), ("DCL_GEO_SMALL", "dcl_debug_geometry_small_batch"), ("DCL_CONV_FEW_CAP", "dcl_debug_conv_few_cap"), ("DCL_CONV_FEW_HINT", "dcl_debug_conv_few_hint"), ("DCL_CONV_SLOTS", "dcl_debug_conv_slots"), ("DCL_CONV_ORDER", "dcl_debug_conv_order_mode"), ("DCL_ORDER_LEVEL", "dcl_debug_order_min_level"),
                    ("DCL_ATTN_SPLIT", "dcl_debug_attention_split"), ("DCL_NN_GRID", "dcl_debug_three_nn_grid"), ("DCL_NN_BATCHED", "dcl_debug_nn_batched_mode"), ("DCL_NN_QPT", "dcl_debug_nn_qpt"),
                    ("DCL_ATTN_XCD", "dcl_debug_attention_xcd_remap"), ("DCL_ATTN_VARIANT", "dcl_debug_attention_variant")):
        if os.environ.get(env):
            getattr(L, fn)(int(os.environ[env]))
    return L
