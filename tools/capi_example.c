/* capi_example.c -- the drop-in boundary from plain C99: include/zkmle.h + libzkmle_amd.so, no C++ / Python / torch.
 * Reproduces the reference's first known answer (polynomials/src/multilinear/evaluation_form.rs:180-185):
 *   partial_evaluate([0,0,3,8], variable 0, value 6) == [18,48]   over ark_bn254::Fq
 * and evaluate([0,0,3,8], [6,2]) == 78 (:213-220).  Exit code 0 = ok (or no GPU present: nothing to compute). */
#include <stdio.h>
#include <string.h>

#include "../include/zkmle.h"

static int check(int rc, const char *what) {
    if (rc != ZK_OK) fprintf(stderr, "%s: %s (%s)\n", what, zk_status_message(rc), zk_last_error());
    return rc;
}

int main(void) {
    int ndev = 0;
    printf("%s\n", zk_version());
    zk_device_count(&ndev);
    if (ndev == 0) {
        uint64_t dummy[4] = {0}, out[8];
        int rc = zk_host_partial_evaluate(ZK_BN254_FQ, dummy, 4, 0, dummy, out);
        printf("no HIP device: compute entry points return %d (%s)\n", rc, zk_status_message(rc));
        return rc == ZK_E_NO_DEVICE ? 0 : 1;
    }
    if (check(zk_init(0), "zk_init")) return 1;
    uint64_t canon[4 * 4] = {0}, table[4 * 4], r[4], point[2 * 4], folded[2 * 4], back[2 * 4], ev[4], evc[4];
    canon[2 * 4] = 3; canon[3 * 4] = 8;                                  /* [0, 0, 3, 8] */
    if (check(zk_vec_from_canonical(ZK_BN254_FQ, canon, 4, table), "from_canonical")) return 1;
    zk_fe_from_u64(ZK_BN254_FQ, 6, r);
    zk_fe_from_u64(ZK_BN254_FQ, 6, point);
    zk_fe_from_u64(ZK_BN254_FQ, 2, point + 4);
    if (check(zk_host_partial_evaluate(ZK_BN254_FQ, table, 4, 0, r, folded), "partial_evaluate")) return 1;
    zk_vec_to_canonical(ZK_BN254_FQ, folded, 2, back);
    if (check(zk_host_evaluate(ZK_BN254_FQ, table, 4, point, 2, ev), "evaluate")) return 1;
    zk_vec_to_canonical(ZK_BN254_FQ, ev, 1, evc);
    printf("partial_evaluate -> [%llu, %llu], evaluate -> %llu\n", (unsigned long long)back[0], (unsigned long long)back[4],
           (unsigned long long)evc[0]);
    /* the reference's panic on a non power-of-two table (evaluation_form.rs:171-176) is a status, not an abort */
    zk_table *t = NULL;
    int rc = zk_table_upload(ZK_BN254_FQ, table, 3, &t);
    printf("upload of 3 entries -> %d (%s)\n", rc, zk_status_message(rc));
    return (back[0] == 18 && back[4] == 48 && evc[0] == 78 && rc == ZK_E_NOT_POW2) ? 0 : 1;
}
