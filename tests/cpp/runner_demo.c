/* The reference runners' call sequence (realesrgan/mtk/cpp/jni/src/main.cpp:42-109: init -> inference_ex ->
 * free -> release) from plain C against include/realesrgan.h, edsr.h and rcan.h, so the context structs are
 * the headers' own and not a re-typed copy.
 *   runner_demo realesrgan|edsr|rcan model.srmi in.u8 W H out.u8          one inference
 *   runner_demo sr model.srmi in.u8 W H out.u8                             the generic runner of include/sr.h (any network, native size)
 *   runner_demo threads modelA.srmi modelB.srmi in.u8 W H outA.u8 outB.u8  two Real-ESRGAN contexts, one per
 *                                                                           thread, three inferences each at once
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "edsr.h"
#include "rcan.h"
#include "realesrgan.h"
#include "sr.h"

static unsigned char* read_file(const char* path, size_t n) {
    unsigned char* p = (unsigned char*)malloc(n);
    FILE* f = fopen(path, "rb");
    if (!p || !f || fread(p, 1, n, f) != n) { fprintf(stderr, "cannot read %zu bytes of %s\n", n, path); exit(1); }
    fclose(f);
    return p;
}

static void write_file(const char* path, const unsigned char* p, size_t n) {
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, n, f) != n) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
    fclose(f);
}

struct job {
    const char* model;
    const unsigned char* img;
    int w, h;
    const char* out_path;
    int rc;
};

static void* thread_main(void* arg) {
    struct job* j = (struct job*)arg;
    realesrgan_context_t ctx;
    j->rc = 1;
    if (realesrgan_init(j->model, &ctx) != 0) return NULL;
    unsigned char* first = NULL;
    size_t bytes = 0;
    for (int rep = 0; rep < 3; ++rep) {
        unsigned char* out = NULL;
        int ow = 0, oh = 0;
        if (realesrgan_inference_ex(&ctx, j->img, j->w, j->h, &out, &ow, &oh) != 0) return NULL;
        bytes = (size_t)ow * oh * 3;
        if (!first) first = out;
        else {
            if (memcmp(first, out, bytes) != 0) { fprintf(stderr, "repeat %d differs\n", rep); return NULL; }
            free(out);
        }
    }
    write_file(j->out_path, first, bytes);
    free(first);
    if (realesrgan_release(&ctx) != 0) return NULL;
    j->rc = 0;
    return NULL;
}

int main(int argc, char** argv) {
    if (argc == 9 && !strcmp(argv[1], "threads")) {
        const int w = atoi(argv[5]), h = atoi(argv[6]);
        unsigned char* img = read_file(argv[4], (size_t)w * h * 3);
        struct job jobs[2] = {{argv[2], img, w, h, argv[7], 1}, {argv[3], img, w, h, argv[8], 1}};
        pthread_t t[2];
        for (int i = 0; i < 2; ++i) pthread_create(&t[i], NULL, thread_main, &jobs[i]);
        for (int i = 0; i < 2; ++i) pthread_join(t[i], NULL);
        free(img);
        if (jobs[0].rc || jobs[1].rc) return 1;
        printf("ok threads\n");
        return 0;
    }
    if (argc != 7) {
        fprintf(stderr, "usage: %s realesrgan|edsr|rcan|sr model.srmi in.u8 W H out.u8\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[4]), h = atoi(argv[5]);
    unsigned char* img = read_file(argv[3], (size_t)w * h * 3);
    unsigned char* out = NULL;
    int ow = 0, oh = 0;
    if (!strcmp(argv[1], "realesrgan")) {
        realesrgan_context_t ctx;
        /* an uninitialised context: inference and release fail (realesrgan.cpp:197-199,222-226) */
        memset(&ctx, 0, sizeof(ctx));
        if (realesrgan_inference_ex(&ctx, img, w, h, &out, &ow, &oh) != -1 || realesrgan_release(&ctx) != -1) return 1;
        if (realesrgan_init(argv[2], &ctx) != 0) return 1;
        if (realesrgan_inference_ex(&ctx, img, w, h, &out, &ow, &oh) != 0) return 1;
        if (ow != ctx.output_width || oh != ctx.output_height) return 1;
        write_file(argv[6], out, (size_t)ow * oh * 3);
        free(out);
        if (realesrgan_release(&ctx) != 0 || ctx.initialized) return 1;
    } else if (!strcmp(argv[1], "sr")) {
        sr_context_t ctx;
        memset(&ctx, 0, sizeof(ctx));
        if (sr_release(&ctx) != 0 || sr_inference(&ctx, img, w, h, &out, &ow, &oh) != -1) return 1;
        if (sr_init(argv[2], &ctx) != 0) { fprintf(stderr, "%s\n", sr_last_error()); return 1; }
        if (sr_reserve(&ctx, w, h) != 0) return 1;
        if (sr_inference(&ctx, img, w, h, &out, &ow, &oh) != 0) return 1;
        if (ow != w * ctx.scale_factor || oh != h * ctx.scale_factor || ctx.channels != 3) return 1;
        write_file(argv[6], out, (size_t)ow * oh * 3);
        free(out);
        if (sr_release(&ctx) != 0 || ctx.initialized) return 1;
    } else if (!strcmp(argv[1], "edsr")) {
        edsr_context_t ctx;
        memset(&ctx, 0, sizeof(ctx));
        if (edsr_release(&ctx) != 0) return 1; /* edsr.cpp:185-187 */
        if (edsr_init(argv[2], &ctx) != 0) return 1;
        if (edsr_inference_ex(&ctx, img, w, h, &out, NULL, NULL) != 0) return 1; /* NULL sizes allowed, edsr.cpp:293-294 */
        write_file(argv[6], out, (size_t)ctx.output_width * ctx.output_height * 3);
        free(out);
        if (edsr_release(&ctx) != 0) return 1;
    } else {
        rcan_context_t ctx;
        if (rcan_init(argv[2], &ctx) != 0) return 1;
        if (rcan_inference_ex(&ctx, img, w, h, &out, &ow, &oh) != 0) return 1;
        write_file(argv[6], out, (size_t)ow * oh * 3);
        free(out);
        if (rcan_release(&ctx) != 0) return 1;
    }
    free(img);
    printf("ok %dx%d\n", ow, oh);
    return 0;
}
