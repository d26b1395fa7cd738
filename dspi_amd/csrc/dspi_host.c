/*
 * dspi_host.c — the thin C host of the DSPi chain on MI355X.
 *
 * Plays the role of the firmware's main loop + USB front end (firmware/DSPi/main.c:736-743,
 * usb_audio.c:1326-1332) for many streams at once: loads a DSPi preset slot or bulk-parameter
 * blob from a file, feeds interleaved PCM through dspi_process() in packets, prints throughput
 * and the status block of stream 0.  Everything DSP happens behind the C-ABI (include/dspi.h).
 *
 *   dspi_host [-f q28|f32|f32fma] [-s streams] [-r rate] [-b block_len] [-n blocks] [-c calls]
 *             [-B bulk.bin | -P slot.bin] [-i pcm16le.raw] [-o pairs.raw] [-v volume_db]
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dspi.h"

static void *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc((size_t)n);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { perror("fread"); exit(2); }
    fclose(f);
    *len = (size_t)n;
    return p;
}

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    int flavor = DSPI_FLAVOR_RP2350_F32;
    uint32_t streams = 4096, rate = 48000, block_len = 48, blocks = 100, calls = 10;
    const char *bulk = NULL, *slot = NULL, *in = NULL, *outp = NULL;
    double vol_db = 0.0;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i], *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "-f") && v) {      /* f32fma: the float flavour with the firmware build's fused multiply-adds (dspi.h) */
            flavor = !strcmp(v, "q28") ? DSPI_FLAVOR_RP2040_Q28 : (!strcmp(v, "f32fma") ? DSPI_FLAVOR_RP2350_F32_FMA : DSPI_FLAVOR_RP2350_F32); i++;
        }
        else if (!strcmp(a, "-s") && v) { streams = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-r") && v) { rate = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-b") && v) { block_len = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-n") && v) { blocks = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-c") && v) { calls = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-B") && v) { bulk = v; i++; }
        else if (!strcmp(a, "-P") && v) { slot = v; i++; }
        else if (!strcmp(a, "-i") && v) { in = v; i++; }
        else if (!strcmp(a, "-o") && v) { outp = v; i++; }
        else if (!strcmp(a, "-v") && v) { vol_db = atof(v); i++; }
        else { fprintf(stderr, "usage: %s [-f q28|f32|f32fma] [-s streams] [-r rate] [-b block_len] [-n blocks] [-c calls] [-B bulk.bin|-P slot.bin] [-i pcm.raw] [-o pairs.raw] [-v vol_db]\n", argv[0]); return 2; }
    }
    dspi_ctx *ctx = NULL;
    int rc = dspi_create(&ctx, flavor, streams, 0);
    if (rc) { fprintf(stderr, "dspi_create failed (%d): the HIP library needs a GPU\n", rc); return 1; }
    if ((rc = dspi_set_sample_rate(ctx, DSPI_ALL_STREAMS, rate))) { fprintf(stderr, "rate: %d\n", rc); return 1; }
    dspi_set_host_volume(ctx, DSPI_ALL_STREAMS, (int16_t)lrint(vol_db * 256.0));
    size_t len;
    if (bulk) { void *b = slurp(bulk, &len); rc = dspi_load_bulk(ctx, DSPI_ALL_STREAMS, b, len); printf("bulk_params_apply -> %d\n", rc); free(b); }
    if (slot) { void *b = slurp(slot, &len); rc = dspi_load_preset_slot(ctx, DSPI_ALL_STREAMS, b, len, -1); printf("preset_load -> %d\n", rc); free(b); }

    const size_t frames = (size_t)blocks * block_len;
    int16_t *pcm = (int16_t *)malloc((size_t)streams * frames * 4);
    if (in) {
        size_t n; int16_t *one = (int16_t *)slurp(in, &n);
        size_t have = n / 4;
        for (uint32_t s = 0; s < streams; s++)
            for (size_t f = 0; f < frames; f++) { pcm[(s * frames + f) * 2] = one[(f % have) * 2]; pcm[(s * frames + f) * 2 + 1] = one[(f % have) * 2 + 1]; }
        free(one);
    } else {   /* xorshift32 white noise at -6 dBFS, per-stream seed as in SURVEY.md 8(d) */
        for (uint32_t s = 0; s < streams; s++) {
            uint32_t x = 0x9E3779B9u ^ (s * 2654435761u);
            if (!x) x = 1;
            for (size_t f = 0; f < frames * 2; f++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; pcm[s * frames * 2 + f] = (int16_t)((int)((x >> 16) % 32769u) - 16384); }
        }
    }
    const int pairs_n = dspi_num_pairs(ctx), ch = dspi_num_channels(ctx);
    dspi_out out;
    out.pairs = (int32_t *)malloc((size_t)streams * pairs_n * frames * 8);
    out.sub = (int32_t *)malloc((size_t)streams * frames * 4);
    out.peaks = (uint16_t *)malloc((size_t)streams * blocks * ch * 2);
    /* the first call also faults in the freshly malloc'ed pages of the I/O buffers and the context's staging buffers: timed on its own */
    double t0 = now(), t_first = 0.0;
    for (uint32_t c = 0; c < calls; c++) {
        if ((rc = dspi_process(ctx, pcm, 16, blocks, block_len, &out, 0))) { fprintf(stderr, "dspi_process: %d %s\n", rc, dspi_last_error(ctx)); return 1; }
        if (c == 0) t_first = now() - t0;
    }
    double dt = now() - t0;
    double fps = calls > 1 ? (double)streams * frames * (calls - 1) / (dt - t_first) : (double)streams * frames / dt;
    printf("%u streams x %u packets x %u frames x %u calls: %.3f s (first call %.3f s), %.3e frames/s (host buffers, PCIe copies included), %.0f real-time streams\n",
           streams, blocks, block_len, calls, dt, t_first, fps, fps / rate);
    uint8_t st[64];
    int n = dspi_get_status(ctx, 0, st, sizeof st);
    printf("stream 0 status (%d bytes): peaks", n);
    for (int i = 0; i < ch; i++) printf(" %u", st[i * 2] | (st[i * 2 + 1] << 8));
    printf(" clip 0x%04x\n", st[ch * 2 + 2] | (st[ch * 2 + 3] << 8));
    if (outp) { FILE *f = fopen(outp, "wb"); fwrite(out.pairs, 8, (size_t)pairs_n * frames, f); fclose(f); }
    dspi_destroy(ctx);
    return 0;
}
