/* Fixture generator (test infrastructure, this container only): writes a Keras ``save_weights``-style
 * HDF5 file with the REAL HDF5 library (libhdf5 1.10.6 under /opt/conda), so that the product's pure-Python
 * reader (amt_saga/hdf5.py) is checked against bytes it did not produce.
 *
 *   gcc -I/opt/conda/include keras_h5_writer.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -o /tmp/keras_h5_writer
 *   /tmp/keras_h5_writer manifest.txt weights.bin out.h5
 *
 * Layout, as keras.engine.saving.save_weights_to_hdf5_group writes it (Keras 2.2 / tf.keras 1.13):
 *   /            attrs  layer_names (fixed-length string array, model.layers order), backend, keras_version
 *   /<layer>     attr   weight_names (fixed-length string array; an empty float64 array for weightless layers)
 *   /<layer>/<layer>/<weight>:0   float32 dataset (the weight name contains '/', hence the nested group)
 * manifest.txt:  "layer <name>" | "weight <name> <offset-in-floats> <ndim> <d0> ..." (weights follow their layer)
 */
#include "hdf5.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXN 4096
static char names[MAXN][128];

static void str_array_attr(hid_t obj, const char *attr, char (*vals)[128], int n) {
    if (n == 0) {                                   /* np.asarray([]) -> float64, shape (0,) */
        hsize_t z = 0;
        hid_t sp = H5Screate_simple(1, &z, NULL);
        hid_t a = H5Acreate2(obj, attr, H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT);
        H5Aclose(a); H5Sclose(sp);
        return;
    }
    size_t w = 1;
    for (int i = 0; i < n; ++i) if (strlen(vals[i]) > w) w = strlen(vals[i]);
    char *buf = calloc((size_t)n, w);
    for (int i = 0; i < n; ++i) memcpy(buf + (size_t)i * w, vals[i], strlen(vals[i]));
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, w);
    H5Tset_strpad(t, H5T_STR_NULLPAD);              /* numpy 'S' dtype as h5py maps it */
    hsize_t d = (hsize_t)n;
    hid_t sp = H5Screate_simple(1, &d, NULL);
    hid_t a = H5Acreate2(obj, attr, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, buf);
    H5Aclose(a); H5Sclose(sp); H5Tclose(t); free(buf);
}

static void scalar_str_attr(hid_t obj, const char *attr, const char *val) {
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, strlen(val));
    H5Tset_strpad(t, H5T_STR_NULLPAD);
    hid_t sp = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(obj, attr, t, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(a, t, val);
    H5Aclose(a); H5Sclose(sp); H5Tclose(t);
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    FILE *m = fopen(argv[1], "r"), *wb = fopen(argv[2], "rb");
    if (!m || !wb) return 3;
    fseek(wb, 0, SEEK_END);
    long nb = ftell(wb);
    fseek(wb, 0, SEEK_SET);
    float *w = malloc((size_t)nb);
    if (fread(w, 1, (size_t)nb, wb) != (size_t)nb) return 4;
    hid_t f = H5Fcreate(argv[3], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    hid_t lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(lcpl, 1);
    static char layers[MAXN][128];
    int nl = 0, nw = 0;
    hid_t g = -1;
    char line[1024], kind[16], name[128];
    while (fgets(line, sizeof line, m)) {
        int pos = 0;
        if (sscanf(line, "%15s %127s%n", kind, name, &pos) < 2) continue;
        if (!strcmp(kind, "layer")) {
            if (g >= 0) { str_array_attr(g, "weight_names", names, nw); H5Gclose(g); }
            strcpy(layers[nl++], name);
            g = H5Gcreate2(f, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
            nw = 0;
        } else if (!strcmp(kind, "weight")) {
            long off; int nd; hsize_t dims[8];
            char *p = line + pos;
            int used = 0;
            sscanf(p, "%ld %d%n", &off, &nd, &used); p += used;
            for (int i = 0; i < nd; ++i) { unsigned long long v; sscanf(p, "%llu%n", &v, &used); p += used; dims[i] = v; }
            strcpy(names[nw++], name);
            hid_t sp = nd ? H5Screate_simple(nd, dims, NULL) : H5Screate(H5S_SCALAR);
            hid_t ds = H5Dcreate2(g, name, H5T_IEEE_F32LE, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT);
            H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, w + off);
            H5Dclose(ds); H5Sclose(sp);
        }
    }
    if (g >= 0) { str_array_attr(g, "weight_names", names, nw); H5Gclose(g); }
    str_array_attr(f, "layer_names", layers, nl);
    scalar_str_attr(f, "backend", "tensorflow");
    scalar_str_attr(f, "keras_version", "2.2.4-tf");
    H5Pclose(lcpl);
    H5Fclose(f);
    return 0;
}
