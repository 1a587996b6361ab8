/* A plain-C consumer of the C ABI (tests/test_host.py builds it with gcc -std=c99 and runs it): the three public headers compile as
 * C, the library links, and the host-side entry points -- no device needed -- behave as documented: error codes instead of exit(),
 * the batch split, the split of a gathered record buffer. */
#include "yolo_hip.h"
#include "yolo_dist.h"
#include "darknet_hip.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)

int main(void)
{
    char err[256] = "";
    CHECK(yolo_create(NULL, err, sizeof err) == NULL && strstr(err, "yolo_config"));
    CHECK(yolo_op_conv_num_cfgs() > 40);

    /* world 3, batch 8: 3 + 3 + 2 */
    int first = -1, count = -1, sum = 0;
    for (int r = 0; r < 3; ++r) {
        CHECK(yolo_shard_bounds(8, 3, r, &first, &count) == YOLO_OK && first == sum);
        sum += count;
    }
    CHECK(sum == 8 && count == 2);
    CHECK(yolo_shard_bounds(8, 3, 3, &first, &count) == YOLO_ERR_INVALID);

    /* what three ranks' all-gather leaves: [3][per * max_out records | per counts], per = 3, max_out = 2; rank 2 serves two images */
    enum { WORLD = 3, BATCH = 8, PER = 3, MAXO = 2 };
    const size_t flat = yolo_dist_flat_words(PER, MAXO);
    CHECK(flat == PER * MAXO * 6 + PER);
    int32_t *g = (int32_t *)calloc(WORLD * flat, sizeof(int32_t));
    for (int r = 0; r < WORLD; ++r) {
        yolo_box *b = (yolo_box *)(g + r * flat);
        int32_t *c = g + r * flat + PER * MAXO * 6;
        for (int i = 0; i < PER; ++i) {
            c[i] = 1 + (r + i) % 2;
            for (int k = 0; k < MAXO; ++k) { b[i * MAXO + k].score = (float)(100 * r + 10 * i + k); b[i * MAXO + k].cls = r; }
        }
    }
    yolo_box boxes[BATCH * MAXO]; int32_t counts[BATCH];
    CHECK(yolo_dist_split_records(g, WORLD, BATCH, MAXO, boxes, counts) == YOLO_OK);
    for (int img = 0; img < BATCH; ++img) {
        const int r = img / 3, i = img % 3;          /* 3 + 3 + 2 */
        CHECK(counts[img] == 1 + (r + i) % 2);
        for (int k = 0; k < MAXO; ++k) CHECK(boxes[img * MAXO + k].score == (float)(100 * r + 10 * i + k) && boxes[img * MAXO + k].cls == r);
    }
    CHECK(yolo_dist_split_records(NULL, WORLD, BATCH, MAXO, boxes, counts) == YOLO_ERR_INVALID);
    CHECK(yolo_dist_create(NULL, 1, 0, NULL, NULL, 1, 1, err, sizeof err) == NULL && strstr(err, "context"));
    free(g);
    puts("abi_consumer ok");
    return 0;
}
