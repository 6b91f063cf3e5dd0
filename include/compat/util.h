// util.h -- host utilities with the names the reference's drivers use (reference include/util.h),
// written for HIP.  Header-only; globals are `inline` so several translation units may include it.
//
//   globals n, m, feature_len, GPUNUM, NEINUM, outfea, inputgraph, reorderfile, rows, reverse_rows
//                                                       (reference util.h:39-71, src/util.cu:3-22)
//   argParse                                            (reference src/util.cu:24-147, same flag names)
//   timestamp / getDuration / getFLOP                   (reference util.h:80,114-128)
//   checkHipErrors / FatalError                         (reference util.h:82-104, hip instead of cuda)
//   hipMalloc2 / registerPtr / safeFree / createCopy / copyVec2Dev / CSRSubGraph   (util.h:144-221)
//   dbg(x)                                              (reference vendors dbg-macro; a one-line logger here)
#ifndef GNNAGG_COMPAT_UTIL_H
#define GNNAGG_COMPAT_UTIL_H

#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <cassert>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../gnnagg.h"

#define CEIL(a, b) (((a) + (b)-1) / (b))

inline int GPUNUM = 1;
inline int NEINUM = -1;
inline int n = 0, m = 0, feature_len = 0;
inline int outfea = 0;
inline long total_size = 0;
inline int *rows = nullptr;
inline int *reverse_rows = nullptr;
inline std::vector<void *> registered_ptr;
inline std::string inputgraph, reorderfile, partitionfile, datadir_global = "../data/", reorder_suffix_global;
inline int **gptrs = nullptr, **gidxs = nullptr;

#define dbg(x) (std::cerr << "[" << __FILE__ << ":" << __LINE__ << "] " << #x << " = " << (x) << std::endl)

inline bool fexist(const std::string &name)
{
    struct stat buffer;
    return stat(name.c_str(), &buffer) == 0;
}

#define timestamp(__var__) auto __var__ = std::chrono::system_clock::now();

inline double getDuration(std::chrono::time_point<std::chrono::system_clock> a,
                          std::chrono::time_point<std::chrono::system_clock> b)
{
    return std::chrono::duration<double>(b - a).count();
}

inline double getFLOP(double time)
{
    assert(time > 0 && m > 0 && feature_len > 0);
    return 2.0 * (double)m * (double)feature_len / time / 1e9;  // reference util.h:125 overflows in int; fixed
}

#define FatalError(s)                                                                  \
    do {                                                                               \
        std::cerr << std::string(s) << "\n" << __FILE__ << ':' << __LINE__ << "\nAborting...\n"; \
        (void)hipDeviceReset();                                                        \
        exit(1);                                                                       \
    } while (0)

#define checkHipErrors(status)                                                         \
    do {                                                                               \
        hipError_t _st = (status);                                                     \
        if (_st != hipSuccess) {                                                       \
            std::stringstream _error;                                                  \
            _error << "Hip failure: " << hipGetErrorString(_st);                       \
            FatalError(_error.str());                                                  \
        }                                                                              \
    } while (0)

// status codes of the C-ABI (include/gnnagg.h), same abort-on-error policy
#define checkGnnagg(status)                                                            \
    do {                                                                               \
        if ((status) != GNNAGG_OK) FatalError(std::string("gnnagg failure: ") + gnnagg_last_error()); \
    } while (0)

inline hipError_t hipMalloc2(void **a, size_t s)
{
    if (s == 0) return hipSuccess;
    total_size += (long)s;
    return hipMalloc(a, ((s + 511) / 512) * 512);
}

template <class T>
inline void registerPtr(T ptr)
{
    registered_ptr.push_back((void *)(ptr));
}

template <class T>
void safeFree(T *&a)
{
    for (auto item : registered_ptr)
        if ((void *)a == item) return;
    if (a != nullptr) {
        (void)hipFree(a);
        (void)hipGetLastError();
        a = nullptr;
    }
}

template <class T>
T *createCopy(T *p, int size)
{
    T *p_d = nullptr;
    checkHipErrors(hipMalloc2((void **)&p_d, size * sizeof(T)));
    checkHipErrors(hipMemcpy(p_d, p, sizeof(T) * size, hipMemcpyHostToDevice));
    return p_d;
}

template <class T>
void copyVec2Dev(std::vector<T> *vec, T *&output)
{
    assert(output == nullptr);
    checkHipErrors(hipMalloc2((void **)&output, vec->size() * sizeof(T)));
    checkHipErrors(hipMemcpy(output, vec->data(), vec->size() * sizeof(T), hipMemcpyHostToDevice));
    std::vector<T>().swap(*vec);
}

class CSRSubGraph
{
public:
    CSRSubGraph(int *outvertexset, int *outptr, int *outidx, int vertex_num, int edge_num)
        : vertexset(outvertexset), ptr(outptr), idx(outidx), num_v(vertex_num), num_e(edge_num) {}
    void free()
    {
        safeFree(vertexset);
        safeFree(ptr);
        safeFree(idx);
    }
    int *vertexset = nullptr;
    int *ptr = nullptr;
    int *idx = nullptr;
    int num_v = 0;
    int num_e = 0;
};

// GNNAGG_COMPAT_DUMP=<dir> (or `--dump <dir>` of the drivers under drivers/): the class shim writes the operands of the LAST call of each
// of its entry points as raw little-endian arrays <dir>/<entry point>.<operand>.bin -- device memory copied out behind a device
// synchronise -- so that a driver's NUMBERS, not only its exit code, can be checked from outside (tests/test_gpu_reference.py,
// tests/test_gpu_parity.py compare them with the oracle).  The reference's drivers fill their inputs with cuRAND / hipRAND, so the inputs
// are dumped with the outputs.  Off (one cached getenv) unless the variable is set; timings of a dumping run mean nothing.
inline const char *compat_dump_dir()
{
    static const char *d = getenv("GNNAGG_COMPAT_DUMP");
    return d && *d ? d : nullptr;
}
inline void compat_dump(const char *entry, const char *operand, const void *dev, size_t bytes)
{
    const char *d = compat_dump_dir();
    if (!d || !dev) return;
    std::vector<char> h(bytes);
    checkHipErrors(hipDeviceSynchronize());
    if (bytes) checkHipErrors(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost));
    const std::string path = std::string(d) + "/" + entry + "." + operand + ".bin";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f || fwrite(h.data(), 1, bytes, f) != bytes) FatalError("GNNAGG_COMPAT_DUMP: cannot write " + path);
    fclose(f);
}

// --dataset D --feature-len F [--datadir DIR] [--reorder SUFFIX] [--nei N] [--gpu-num N] [--outfea N]
// [--partition-path P] [--limit N] [--limit2 N]; "--flag value" and "--flag=value" both accepted.
inline void argParse(int argc, char **argv, int *p_limit = nullptr, int *p_limit2 = nullptr)
{
    std::string dset;
    bool have_feat = false, have_reorder = false, have_l1 = false, have_l2 = false;
    std::string reorder_suffix;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], v;
        if (a.rfind("--", 0) != 0) {
            std::cerr << "unexpected argument " << a << std::endl;
            exit(1);
        }
        size_t eq = a.find('=');
        if (eq != std::string::npos) {
            v = a.substr(eq + 1);
            a = a.substr(0, eq);
        } else if (a == "--help") {
            std::cout << "flags: --dataset --datadir --partition-path --reorder --gpu-num --nei --feature-len --outfea "
                         "--limit --limit2\n";
            exit(0);
        } else {
            if (i + 1 >= argc) {
                std::cerr << "flag " << a << " needs a value" << std::endl;
                exit(1);
            }
            v = argv[++i];
        }
        if (a == "--dataset") dset = v;
        else if (a == "--datadir") datadir_global = v;
        else if (a == "--partition-path") partitionfile = v;
        else if (a == "--reorder") { reorder_suffix = v; have_reorder = true; }
        else if (a == "--gpu-num") GPUNUM = atoi(v.c_str());
        else if (a == "--nei") NEINUM = atoi(v.c_str());
        else if (a == "--feature-len") { feature_len = atoi(v.c_str()); have_feat = true; }
        else if (a == "--outfea") outfea = atoi(v.c_str());
        else if (a == "--limit") { if (p_limit) *p_limit = atoi(v.c_str()); have_l1 = true; }
        else if (a == "--limit2") { if (p_limit2) *p_limit2 = atoi(v.c_str()); have_l2 = true; }
        else {
            std::cerr << "unknown flag " << a << std::endl;
            exit(1);
        }
    }
    if (dset.empty()) FatalError("--dataset is required (reference util.cu:68)");
    if (!have_feat) FatalError("--feature-len is required (reference util.cu:92)");
    if (!datadir_global.empty() && datadir_global.back() != '/') datadir_global += '/';
    const std::string configpath = datadir_global + dset + ".config";
    if (!fexist(configpath)) FatalError("missing " + configpath);
    FILE *fin = fopen(configpath.c_str(), "r");
    if (fscanf(fin, "%d", &n) != 1 || fscanf(fin, "%d", &m) != 1) FatalError("malformed " + configpath);
    fclose(fin);
    if (have_reorder) {
        reorderfile = datadir_global + dset + ".reorder" + (reorder_suffix.size() > 1 ? reorder_suffix : "");
        if (!fexist(reorderfile)) FatalError("missing reorder file " + reorderfile);
        reorder_suffix_global = reorder_suffix.size() > 1 ? reorder_suffix : "";
    } else {
        reorderfile = "";
    }
    if (!partitionfile.empty() && !fexist(partitionfile)) FatalError("missing " + partitionfile);
    if (p_limit && !have_l1) FatalError("--limit is required");
    if (p_limit2 && !have_l2) FatalError("--limit2 is required");
    inputgraph = dset;  // reference util.cu:133: later passed to load_graph as the dataset name
}

// ---- The reference's OWN helper names, spelled as its drivers spell them (reference util.h:29,82-104,144-151, util.cu:157-175).
// With these, a reference driver (Figure8/9/10 main*.cu) needs nothing but the mechanical cuda* -> hip* rename of ROCm's
// hipify-perl to build against this directory and libgnnagg.so -- no hand edits: drivers/build_reference_drivers.sh.
#if __has_include(<hiprand.h>)
#include <hiprand.h>
#endif
#if __has_include(<hipblas.h>)
#include <hipblas.h>
inline hipblasHandle_t cublasH = nullptr;
inline hipblasHandle_t *cublasHs = new hipblasHandle_t[1]();
#endif
#define checkCudaErrors(status)                                                        \
    do {                                                                               \
        if ((int)(status) != 0) {                                                      \
            std::stringstream _error;                                                  \
            _error << "Cuda failure: " << (int)(status);                               \
            FatalError(_error.str());                                                  \
        }                                                                              \
    } while (0)
inline hipError_t cudaMalloc2(void **a, size_t s) { return hipMalloc2(a, s); }
using namespace std;  // reference util.h:29 (its drivers write vector<...>, string, cout unqualified)

#endif
