// flat_cxx_linkage.cpp -- the ten flat functions of Figure7/kernel.cpp:15-35 with C++ linkage, forwarding to the extern "C"
// symbols of libgnnagg.so (include/gnnagg.h section A).  The reference's torch binding DECLARES them without extern "C"
// (its definitions live in Figure7/kernel_generated.cu); with this translation unit in the link, Figure7/kernel.cpp builds
// against the library without an edit (drivers/build_reference_torch_ext.py).  A C++ function and a C function of the same
// name and parameters cannot be declared in one scope, hence the asm labels.
#include <cstdint>
#define C_SYM(ret, name, params) extern "C" ret c_##name params __asm__(#name)
C_SYM(int64_t, GCN_init_impl, (int *, int *, float *, int, int));
C_SYM(void, GCN_update_val_impl, (int64_t, float *));
C_SYM(void, GCN_run_impl, (int64_t, float *, float *, int, int, int));
C_SYM(void, GCN_schedule_impl, (int64_t, int *));
C_SYM(int64_t, GAT_init_impl, (int *, int *, int, int));
C_SYM(void, GAT_run_impl, (int64_t, float *, float *, float *, int, int, int));
C_SYM(void, GAT_run_u_add_v_impl, (int64_t, float *, float *, int));
C_SYM(void, GAT_run_add_to_center_impl, (int64_t, float *, float *, int));
C_SYM(void, GAT_run_div_each_impl, (int64_t, float *, float *, int));
C_SYM(void, GAT_schedule_impl, (int64_t, int *));
int64_t GCN_init_impl(int *p, int *i, float *v, int nv, int ne) { return c_GCN_init_impl(p, i, v, nv, ne); }
void GCN_update_val_impl(int64_t at, float *v) { c_GCN_update_val_impl(at, v); }
void GCN_run_impl(int64_t at, float *f, float *o, int b, int s, int fl) { c_GCN_run_impl(at, f, o, b, s, fl); }
void GCN_schedule_impl(int64_t at, int *a) { c_GCN_schedule_impl(at, a); }
int64_t GAT_init_impl(int *p, int *i, int nv, int ne) { return c_GAT_init_impl(p, i, nv, ne); }
void GAT_run_impl(int64_t at, float *f, float *a, float *o, int b, int s, int fl) { c_GAT_run_impl(at, f, a, o, b, s, fl); }
void GAT_run_u_add_v_impl(int64_t at, float *a, float *o, int b) { c_GAT_run_u_add_v_impl(at, a, o, b); }
void GAT_run_add_to_center_impl(int64_t at, float *i, float *o, int b) { c_GAT_run_add_to_center_impl(at, i, o, b); }
void GAT_run_div_each_impl(int64_t at, float *i, float *o, int b) { c_GAT_run_div_each_impl(at, i, o, b); }
void GAT_schedule_impl(int64_t at, int *a) { c_GAT_schedule_impl(at, a); }
