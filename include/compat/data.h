// data.h -- graph loading with the reference's signature (reference include/data.h:46, src/data.cu:31-139),
// forwarding to the C-ABI loader / reorder (gnnagg_load_graph, gnnagg_reorder_csr).
// Unlike the reference (data.cu:34 hard-codes "../data/") the --datadir of argParse is honoured.
#ifndef GNNAGG_COMPAT_DATA_H
#define GNNAGG_COMPAT_DATA_H
#include "util.h"

template <class T>
T *createCudaMatrixCopy(T *d, int nelem)  // reference data.h:8-16 (name kept; allocates HIP memory)
{
    return createCopy(d, nelem);
}

// indptr / indices come back as new[] host arrays owned by the caller, like the reference.  The reorder
// file is <datadir><dset>.reorder<reorder_subfix> when a suffix is given (data.cu:95-96), else the global
// `reorderfile` set by argParse --reorder (util.cu:101-119); it is applied when shuffle is true and the file
// exists (data.cu:97), filling the globals rows / reverse_rows (data.cu:105-113).
inline void load_graph(std::string dset, int &num_v, int &num_e, int *&indptr, int *&indices, bool shuffle = true,
                       std::string reorder_subfix = "")
{
    int *p = nullptr, *i = nullptr;
    checkGnnagg(gnnagg_load_graph(datadir_global.c_str(), dset.c_str(), "", 0, &num_v, &num_e, &p, &i, nullptr, nullptr));
    if (!reorder_subfix.empty()) reorderfile = datadir_global + dset + ".reorder" + reorder_subfix;
    indptr = new int[num_v + 1];
    indices = new int[num_e > 0 ? num_e : 1];
    if (shuffle && reorderfile.size() > 1 && fexist(reorderfile)) {
        FILE *f = fopen(reorderfile.c_str(), "r");
        rows = new int[num_v > 0 ? num_v : 1];
        reverse_rows = new int[num_v > 0 ? num_v : 1];
        for (int k = 0; k < num_v; ++k) {
            if (fscanf(f, "%d", &rows[k]) != 1 || rows[k] < 0 || rows[k] >= num_v) FatalError("malformed " + reorderfile);
            reverse_rows[rows[k]] = k;
        }
        fclose(f);
        checkGnnagg(gnnagg_reorder_csr(p, i, rows, reverse_rows, num_v, num_e, indptr, indices));
    } else {
        memcpy(indptr, p, sizeof(int) * ((size_t)num_v + 1));
        if (num_e > 0) memcpy(indices, i, sizeof(int) * (size_t)num_e);
    }
    gnnagg_free_host(p);
    gnnagg_free_host(i);
}
#endif
