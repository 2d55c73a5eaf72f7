
#include <cstdio>
#include <vector>
#include "svd3_var.h"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f) / 36; fseek(f, 0, SEEK_SET);
    std::vector<float> in(n * 9), out(n * 21), dbg(n * 14);
    fread(in.data(), 4, n * 9, f); fclose(f);
    for (long i = 0; i < n; ++i) { hps::gesdd3::g_dbg = &dbg[i * 14]; hps::gesdd3::svd3(&in[i * 9], &out[i * 21], &out[i * 21 + 9], &out[i * 21 + 12]); }
    f = fopen("dbg.bin", "wb"); fwrite(dbg.data(), 4, n * 14, f); fclose(f);
    f = fopen(argv[2], "wb"); fwrite(out.data(), 4, n * 21, f); fclose(f);
    return 0;
}
