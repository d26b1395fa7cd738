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
 *   dspi_host -rt ...   the drop-in call as INTEGRATION.md section 3 writes it (the body of usb_audio_drain_ring, usb_audio.c:1326-1332):
 *             ONE packet of block_len frames per dspi_process(), host buffers, `calls` calls back to back, the input advancing through
 *             the PCM file (wrapping); prints the latency per call (p50 / p99 / max) and the sustained rate against real time.
 *             -O all.raw: every call's words of streams 0 and streams-1 (pairs, sub, peaks per call) for the parity check of
 *             tools/bench_realtime.py; -L lat.f64: every call's latency in seconds.
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
static int cmp_d(const void *a, const void *b) { double x = *(const double *)a, y = *(const double *)b; return x < y ? -1 : x > y; }

/* -rt: one packet per call (the firmware's own rhythm: a 1 ms USB packet, usb_audio.c:1326-1332), latency per call */
static int realtime(dspi_ctx *ctx, uint32_t streams, uint32_t rate, uint32_t block_len, uint32_t calls, const int16_t *one, size_t have_frames,
                    const char *all_path, const char *lat_path) {
    const int pairs_n = dspi_num_pairs(ctx), ch = dspi_num_channels(ctx);
    const size_t have_packets = have_frames / block_len;
    if (!have_packets) { fprintf(stderr, "-rt needs at least one packet of input\n"); return 2; }
    int16_t *pcm = (int16_t *)malloc((size_t)streams * block_len * 4);
    dspi_out out;
    memset(&out, 0, sizeof out);
    out.pairs = (int32_t *)malloc((size_t)streams * pairs_n * block_len * 8);
    out.sub = (int32_t *)malloc((size_t)streams * block_len * 4);
    out.peaks = (uint16_t *)malloc((size_t)streams * ch * 2);
    double *lat = (double *)malloc(sizeof(double) * calls);
    FILE *fa = all_path ? fopen(all_path, "wb") : NULL;
    const uint32_t watch[2] = {0, streams - 1};
    int rc;
    /* (the first calls allocate the context's staging buffers and build the launch lists: they are processed like every other packet — the
     *  checker must see exactly the same packets — and reported apart from the steady state) */
    for (uint32_t c = 0; c < calls; c++) {
        const size_t k = c % have_packets;
        /* every stream plays the file, stream s s packets behind (streams differ; a stream's history is a function of the call number alone) */
        for (uint32_t s = 0; s < streams; s++)
            memcpy(pcm + (size_t)s * block_len * 2, one + ((k + have_packets - (s % have_packets)) % have_packets) * block_len * 2, (size_t)block_len * 4);
        const double t0 = now();
        rc = dspi_process(ctx, pcm, 16, 1, block_len, &out, 0);
        lat[c] = now() - t0;
        if (rc) { fprintf(stderr, "dspi_process: %d %s\n", rc, dspi_last_error(ctx)); return 1; }
        if (fa)
            for (int w = 0; w < (streams > 1 ? 2 : 1); w++) {
                const uint32_t s = watch[w];
                fwrite(out.pairs + (size_t)s * pairs_n * block_len * 2, 8, (size_t)pairs_n * block_len, fa);
                fwrite(out.sub + (size_t)s * block_len, 4, block_len, fa);
                fwrite(out.peaks + (size_t)s * ch, 2, (size_t)ch, fa);
            }
    }
    if (fa) fclose(fa);
    if (lat_path) { FILE *f = fopen(lat_path, "wb"); fwrite(lat, sizeof(double), calls, f); fclose(f); }
    double total = 0.0;
    for (uint32_t c = 0; c < calls; c++) total += lat[c];
    /* the steady state: the first 1 % of the calls (buffer allocation, first launches) are reported apart */
    const uint32_t skip = calls / 100 < calls ? calls / 100 : 0;
    double first_max = 0.0;
    for (uint32_t c = 0; c < skip; c++) if (lat[c] > first_max) first_max = lat[c];
    const uint32_t n = calls - skip;
    double *srt = (double *)malloc(sizeof(double) * n);
    memcpy(srt, lat + skip, sizeof(double) * n);
    qsort(srt, n, sizeof(double), cmp_d);
    const double packet_s = (double)block_len / rate;
    printf("rt: %u streams x %u calls of one %u-frame packet (%.0f us of audio): p50 %.1f us  p99 %.1f us  p99.9 %.1f us  max %.1f us  (first %u calls: max %.1f us)  mean %.1f us = %.1f x real time\n",
           streams, calls, block_len, packet_s * 1e6, srt[n / 2] * 1e6, srt[(size_t)(n * 0.99)] * 1e6, srt[(size_t)(n * 0.999)] * 1e6, srt[n - 1] * 1e6, skip, first_max * 1e6,
           total / calls * 1e6, packet_s / (total / calls));
    free(srt); free(lat); free(pcm); free(out.pairs); free(out.sub); free(out.peaks);
    return 0;
}

int main(int argc, char **argv) {
    int flavor = DSPI_FLAVOR_RP2350_F32;
    uint32_t streams = 4096, rate = 48000, block_len = 48, blocks = 100, calls = 10;
    const char *bulk = NULL, *slot = NULL, *in = NULL, *outp = NULL, *allp = NULL, *latp = NULL;
    double vol_db = 0.0;
    int rt = 0;
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
        else if (!strcmp(a, "-rt")) rt = 1;
        else if (!strcmp(a, "-O") && v) { allp = v; i++; }
        else if (!strcmp(a, "-L") && v) { latp = v; i++; }
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

    if (rt) {
        size_t n = 0;
        int16_t *one;
        if (in) one = (int16_t *)slurp(in, &n);
        else {      /* 1 000 packets of xorshift32 white noise at -6 dBFS */
            n = (size_t)1000 * block_len * 4;
            one = (int16_t *)malloc(n);
            uint32_t x = 0x9E3779B9u;
            for (size_t f = 0; f < n / 2; f++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; one[f] = (int16_t)((int)((x >> 16) % 32769u) - 16384); }
        }
        rc = realtime(ctx, streams, rate, block_len, calls, one, n / 4, allp, latp);
        free(one);
        dspi_destroy(ctx);
        return rc;
    }
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
