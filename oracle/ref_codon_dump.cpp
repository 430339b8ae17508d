// oracle/_ref builder input: compiles the reference's OWN GeneticCode.h
// (src/commons/GeneticCode.h, included from /root/reference where it lies;
// it needs nothing but the C++ standard library) and dumps its tables so that
// the oracle's restated codon/base tables can be pinned against them.
// TEST INFRASTRUCTURE ONLY.  Output goes to oracle/_ref/ (git-ignored).
#include <cstdio>
#include <string>
#include <vector>
#include "GeneticCode.h"

int main() {
    GeneticCode gc(false);
    printf("# nuc2aa[8][8][8] then nuc2num[8][8][8] (GeneticCode.h:34-193), then nuc2int(atcg[c]) and nuc2int(iRCT[atcg[c]]) for c=0..255\n");
    for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) for (int c = 0; c < 8; c++) printf("%d ", gc.nuc2aa[a][b][c]);
    printf("\n");
    for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) for (int c = 0; c < 8; c++) printf("%d ", gc.nuc2num[a][b][c]);
    printf("\n");
    for (int c = 0; c < 256; c++) printf("%d ", (int)(nuc2int(gc.atcg[c])));
    printf("\n");
    for (int c = 0; c < 256; c++) printf("%d ", (int)(nuc2int(gc.iRCT[(unsigned char)gc.atcg[c]])));
    printf("\n");
    return 0;
}
