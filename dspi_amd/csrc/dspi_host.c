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
 *   dspi_host -g N [-S weak|strong] [-w warmup] ...   the node-level throughput run: ONE process, one context and one
 *             feeder thread per GPU, streams partitioned gpu = stream / ceil(S / N) (stream_range below = dspi_amd/shard.py), buffers
 *             resident on each device, `calls` timed dspi_process() per device between two thread barriers after `warmup` untimed ones,
 *             then ONE collective — ncclAllReduce (RCCL over xGMI) of sum(frames) and max(seconds), 8 bytes each — and one JSON line with
 *             bench.py's keys.  -S weak (default): -s streams PER GPU; strong: -s streams in total.  -o: device 0's first stream, last call.
 *             -D none: no device (contexts are DSPI_DEVICE_NONE, nothing is processed): the partition, the parameter path of every
 *             context and the reduction's host stand-in, for boxes without a GPU (tests/test_host_binary_cpu.py).
 *             Every feeder thread is pinned to the CPUs of its GPU's NUMA node (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/numa_node ->
 *             /sys/devices/system/node/node<k>/cpulist; -D none: node = rank modulo the nodes present) and the line says so ("affinity");
 *             it carries a "roofline" object (algorithmic HBM bytes from the preset's own delays and enables, read back through the vendor
 *             requests 0x79 / 0x73, over the slowest device's time) and, with -o, the last call's pair words of EVERY device's first
 *             stream (<path> for device 0, <path>.dev<k> for the others) plus a checksum of them in "checked_streams".
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "dspi.h"
/* the HIP runtime's C API and RCCL: used by -g only (device buffers; the final all-reduce) */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
/* DSPI_HOST_RCCL (the Makefile sets it when the ROCm install has the RCCL header): without it -g reduces on the host and says so */
#ifdef DSPI_HOST_RCCL
#include <rccl/rccl.h>
#endif

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
    if (!calls) { fprintf(stderr, "-rt needs -c >= 1\n"); return 2; }
    int16_t *pcm = (int16_t *)malloc((size_t)streams * block_len * 4);
    dspi_out out;
    memset(&out, 0, sizeof out);
    out.pairs = (int32_t *)malloc((size_t)streams * pairs_n * block_len * 8);
    out.sub = (int32_t *)malloc((size_t)streams * block_len * 4);
    out.peaks = (uint16_t *)malloc((size_t)streams * ch * 2);
    double *lat = (double *)malloc(sizeof(double) * calls);
    FILE *fa = all_path ? fopen(all_path, "wb") : NULL;
    const uint32_t watch[2] = {0, streams - 1};
    int rc = 0;
    if (all_path && !fa) { perror(all_path); rc = 2; }
    if (!pcm || !out.pairs || !out.sub || !out.peaks || !lat) { fprintf(stderr, "-rt: out of memory\n"); rc = 2; }
    if (rc) goto done;
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
        if (rc) { fprintf(stderr, "dspi_process: %d %s\n", rc, dspi_last_error(ctx)); rc = 1; goto done; }
        if (fa)
            for (int w = 0; w < (streams > 1 ? 2 : 1); w++) {
                const uint32_t s = watch[w];
                fwrite(out.pairs + (size_t)s * pairs_n * block_len * 2, 8, (size_t)pairs_n * block_len, fa);
                fwrite(out.sub + (size_t)s * block_len, 4, block_len, fa);
                fwrite(out.peaks + (size_t)s * ch, 2, (size_t)ch, fa);
            }
    }
    if (lat_path) {
        FILE *f = fopen(lat_path, "wb");
        if (!f) { perror(lat_path); rc = 2; goto done; }
        fwrite(lat, sizeof(double), calls, f); fclose(f);
    }
    {
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
    /* the tail, not only percentiles: calls longer than the packet they carry (a dropout in the firmware's rhythm), a log2 histogram, and the
     * library's own record of the polling path (dspi_debug_direct_stats) */
    {
        uint32_t over = 0, hist[16];
        memset(hist, 0, sizeof hist);
        for (uint32_t c = skip; c < calls; c++) {
            if (lat[c] > packet_s) over++;
            const double us = lat[c] * 1e6;
            int b = 0;
            while (b < 15 && us >= (double)(16u << b)) b++;       /* bucket b: [8 << b, 16 << b) us; bucket 0: < 16 us; bucket 15: >= 262 144 us */
            hist[b]++;
        }
        uint64_t ds[5] = {0, 0, 0, 0, 0};
        (void)dspi_debug_direct_stats(ctx, ds, 5);
        printf("rt-json: {\"calls\": %u, \"steady_calls\": %u, \"packet_us\": %.3f, \"p50_us\": %.2f, \"p99_us\": %.2f, \"p99_9_us\": %.2f, \"p99_99_us\": %.2f, \"max_us\": %.2f, "
               "\"n_over_packet\": %u, \"hist_log2_us\": {\"first_edge_us\": 16, \"counts\": [",
               calls, n, packet_s * 1e6, srt[n / 2] * 1e6, srt[(size_t)(n * 0.99)] * 1e6, srt[(size_t)(n * 0.999)] * 1e6, srt[(size_t)(n * 0.9999)] * 1e6, srt[n - 1] * 1e6, over);
        for (int b = 0; b < 16; b++) printf("%s%u", b ? ", " : "", hist[b]);
        printf("]}, \"direct_path\": {\"calls\": %llu, \"blocking_waits\": %llu, \"max_enqueue_us\": %.2f, \"max_wait_us\": %.2f, \"spin_budget_us\": %.1f}}\n",
               (unsigned long long)ds[0], (unsigned long long)ds[1], ds[2] / 1e3, ds[3] / 1e3, ds[4] / 1e3);
    }
    free(srt);
    }
done:
    if (fa) fclose(fa);
    free(lat); free(pcm); free(out.pairs); free(out.sub); free(out.peaks);
    return rc;
}

/* ---- -g N: the node-level run, one context + one feeder thread per GPU ---- */
/* contiguous ranges, gpu = stream / ceil(S / n): [first, last)  (dspi_amd/shard.py:stream_range) */
static void stream_range(uint32_t rank, uint32_t world, uint32_t total, uint32_t *first, uint32_t *last) {
    const uint32_t per = (total + world - 1) / world;
    const uint64_t f = (uint64_t)rank * per;
    *first = f < total ? (uint32_t)f : total;
    *last = (uint64_t)*first + per < total ? *first + per : total;
}

typedef struct {
    /* in */
    int rank, world, dry, flavor;
    uint32_t first, last, rate, block_len, blocks, calls, warmup;
    double vol_db;
    const void *bulk, *slot; size_t bulk_len, slot_len;
    const char *outp;
    pthread_barrier_t *bar;
    /* out */
    int rc; char err[200];
    double frames, seconds;
    int channels;
    int numa_node, cpus_pinned; char pin_src[48];       /* the feeder thread's affinity */
    double bytes_strict, bytes_resident;                /* algorithmic HBM bytes per frame of this shard's preset */
    uint32_t check_sum, check_words; int checked;       /* the last call's pair words of the shard's first stream */
} Shard;

/* "0-3,8,10-11" -> cpu_set_t; returns the number of CPUs set */
static int parse_cpulist(const char *s, cpu_set_t *set) {
    int n = 0;
    CPU_ZERO(set);
    while (*s) {
        char *e;
        long a = strtol(s, &e, 10), b = a;
        if (e == s) break;
        if (*e == '-') { s = e + 1; b = strtol(s, &e, 10); if (e == s) break; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && !CPU_ISSET((int)c, set)) { CPU_SET((int)c, set); n++; }
        s = e;
        while (*s == ',' || *s == '\n' || *s == ' ') s++;
    }
    return n;
}
static int read_line(const char *path, char *buf, size_t cap) {
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    if (!fgets(buf, (int)cap, f)) { fclose(f); return -1; }
    fclose(f);
    return 0;
}
/* pin the calling thread to the CPUs of the GPU's NUMA node (the feeder's buffers and doorbells stay on the socket
 * the device hangs off).  dry: no device — node = rank modulo the nodes present, so that the parsing and the call run on any box. */
static void pin_to_gpu_node(Shard *h) {
    char path[160], buf[4096];
    int node = -1;
    h->numa_node = -1; h->cpus_pinned = 0; snprintf(h->pin_src, sizeof h->pin_src, "none");
    if (!h->dry) {
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, h->rank) != hipSuccess) return;
        for (char *q = bus; *q; q++) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');      /* sysfs names are lower case */
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
        if (read_line(path, buf, sizeof buf)) return;
        node = atoi(buf);
        snprintf(h->pin_src, sizeof h->pin_src, "pci");
        if (node < 0) { snprintf(h->pin_src, sizeof h->pin_src, "pci: no node"); return; }      /* single-node boxes and VMs report -1: nothing to pin to */
    } else {
        int nodes = 0;
        for (;; nodes++) { snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nodes); if (read_line(path, buf, sizeof buf)) break; }
        if (!nodes) return;
        node = h->rank % nodes;
        snprintf(h->pin_src, sizeof h->pin_src, "dry: rank %% %d nodes", nodes);
    }
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    if (read_line(path, buf, sizeof buf)) return;
    cpu_set_t want, have;
    if (parse_cpulist(buf, &want) < 1) return;
    /* only CPUs this process may use at all (cgroup / taskset): the intersection, if it is not empty */
    if (sched_getaffinity(0, sizeof have, &have) == 0) {
        cpu_set_t both; CPU_AND(&both, &want, &have);
        if (CPU_COUNT(&both) > 0) want = both; else return;
    }
    if (pthread_setaffinity_np(pthread_self(), sizeof want, &want) != 0) return;
    h->numa_node = node; h->cpus_pinned = CPU_COUNT(&want);
}

/* Algorithmic HBM bytes per frame of the context's preset, two of bench.py's three ways (bench.py:algorithmic_bytes): the delays and enables are
 * read back through the firmware's own control requests (REQ_GET_OUTPUT_DELAY 0x79, REQ_GET_OUTPUT_ENABLE 0x73, usb_audio.c vendor handler). */
static void algorithmic_bytes(dspi_ctx *ctx, uint32_t rate, size_t frames_per_call, double *strict, double *resident) {
    const int n_out = dspi_num_outputs(ctx), float_flavor = dspi_num_channels(ctx) == 11;
    const double max_d = float_flavor ? 4096.0 : 2048.0, T = (double)frames_per_call;
    const double io = 4.0 + 4.0 * (n_out - 1) + 4.0;
    *strict = io; *resident = io;
    for (int o = 0; o < n_out; o++) {
        float ms = 0.0f; uint8_t en = 0;
        if (dspi_vendor_get(ctx, 0, 0x79, (uint16_t)o, &ms, 4) != 4 || dspi_vendor_get(ctx, 0, 0x73, (uint16_t)o, &en, 1) != 1) continue;
        double total_ms = (double)ms + (o == n_out - 1 ? 128.0 / rate * 1000.0 : 0.0);      /* the sub's alignment delay */
        double d = floor(total_ms * rate / 1000.0);
        if (d < 0.0) d = 0.0;
        if (d > max_d) d = max_d;
        if (d > 0.0 && en) {
            *resident += 8.0;
            *strict += 4.0 * (d < T ? d : T) / T + 4.0 * (max_d < T ? max_d : T) / T;
        }
    }
}

static void *shard_main(void *arg) {
    Shard *h = (Shard *)arg;
    const uint32_t S = h->last - h->first;
    const size_t frames = (size_t)h->blocks * h->block_len;
    dspi_ctx *ctx = NULL;
    int16_t *d_pcm = NULL; int32_t *d_pairs = NULL, *d_sub = NULL; uint16_t *d_peaks = NULL;
    int in_barrier = 0;
    h->frames = 0.0; h->seconds = 0.0; h->rc = 0;
    pin_to_gpu_node(h);       /* before the context exists: its staging buffers and the runtime's queues are then first touched from this node */
#define SHARD_FAIL(...) do { snprintf(h->err, sizeof h->err, __VA_ARGS__); h->rc = 1; goto out; } while (0)
    if (S) {
        int rc = dspi_create(&ctx, h->flavor, S, h->dry ? DSPI_DEVICE_NONE : h->rank);
        if (rc) SHARD_FAIL("dspi_create on device %d: %d", h->rank, rc);
        h->channels = dspi_num_channels(ctx);
        if ((rc = dspi_set_sample_rate(ctx, DSPI_ALL_STREAMS, h->rate))) SHARD_FAIL("rate: %d", rc);
        dspi_set_host_volume(ctx, DSPI_ALL_STREAMS, (int16_t)lrint(h->vol_db * 256.0));
        if (h->bulk && (rc = dspi_load_bulk(ctx, DSPI_ALL_STREAMS, h->bulk, h->bulk_len))) SHARD_FAIL("bulk_params_apply -> %d", rc);
        if (h->slot && (rc = dspi_load_preset_slot(ctx, DSPI_ALL_STREAMS, h->slot, h->slot_len, -1))) SHARD_FAIL("preset_load -> %d", rc);
        algorithmic_bytes(ctx, h->rate, frames, &h->bytes_strict, &h->bytes_resident);
    }
    if (S && !h->dry) {
        const int pairs_n = dspi_num_pairs(ctx);
        const size_t in_b = (size_t)S * frames * 4, pairs_b = (size_t)S * pairs_n * frames * 8, sub_b = (size_t)S * frames * 4, peaks_b = (size_t)S * h->blocks * h->channels * 2;
        if (hipSetDevice(h->rank) != hipSuccess) SHARD_FAIL("hipSetDevice(%d)", h->rank);
        if (hipMalloc((void **)&d_pcm, in_b) != hipSuccess || hipMalloc((void **)&d_pairs, pairs_b) != hipSuccess ||
            hipMalloc((void **)&d_sub, sub_b) != hipSuccess || hipMalloc((void **)&d_peaks, peaks_b) != hipSuccess) SHARD_FAIL("hipMalloc of %zu MB on device %d", (in_b + pairs_b + sub_b + peaks_b) >> 20, h->rank);
        /* xorshift32 white noise at -6 dBFS, per-stream seed by the GLOBAL stream index, uploaded in slabs of whole streams */
        const uint32_t slab = S < 256 ? S : 256;
        int16_t *buf = (int16_t *)malloc((size_t)slab * frames * 4);
        if (!buf) SHARD_FAIL("out of memory");
        for (uint32_t s0 = 0; s0 < S; s0 += slab) {
            const uint32_t n = S - s0 < slab ? S - s0 : slab;
            for (uint32_t s = 0; s < n; s++) {
                uint32_t x = 0x9E3779B9u ^ ((h->first + s0 + s) * 2654435761u);
                if (!x) x = 1;
                int16_t *q = buf + (size_t)s * frames * 2;
                for (size_t f = 0; f < frames * 2; f++) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; q[f] = (int16_t)((int)((x >> 16) % 32769u) - 16384); }
            }
            if (hipMemcpy(d_pcm + (size_t)s0 * frames * 2, buf, (size_t)n * frames * 4, hipMemcpyHostToDevice) != hipSuccess) { free(buf); SHARD_FAIL("hipMemcpy H2D"); }
        }
        free(buf);
        dspi_out out; memset(&out, 0, sizeof out);
        out.pairs = d_pairs; out.sub = d_sub; out.peaks = d_peaks;
        for (uint32_t c = 0; c < h->warmup; c++) {
            int rc = dspi_process(ctx, d_pcm, 16, h->blocks, h->block_len, &out, DSPI_MEM_DEVICE);
            if (rc) SHARD_FAIL("dspi_process: %d %s", rc, dspi_last_error(ctx));
        }
        if (dspi_sync(ctx)) SHARD_FAIL("dspi_sync: %s", dspi_last_error(ctx));
        in_barrier = 1;
        pthread_barrier_wait(h->bar);                 /* every device idle and warmed up: the timed region starts together */
        const double t0 = now();
        int rc = 0;
        for (uint32_t c = 0; c < h->calls && !rc; c++) rc = dspi_process(ctx, d_pcm, 16, h->blocks, h->block_len, &out, DSPI_MEM_DEVICE);
        if (!rc) rc = dspi_sync(ctx);
        h->seconds = now() - t0;
        pthread_barrier_wait(h->bar);
        in_barrier = 2;
        if (rc) SHARD_FAIL("dspi_process: %d %s", rc, dspi_last_error(ctx));
        h->frames = (double)S * (double)frames * h->calls;
        {                                             /* every device's first stream, the last call: the checker's window on the words */
            const size_t n = (size_t)pairs_n * frames * 8;
            uint32_t *w = (uint32_t *)malloc(n);
            if (!w || hipMemcpy(w, d_pairs, n, hipMemcpyDeviceToHost) != hipSuccess) { free(w); SHARD_FAIL("D2H of the checked stream"); }
            uint32_t sum = 2166136261u;               /* FNV-1a over the words: printed, and recomputed from the oracle's words by the tests */
            for (size_t i = 0; i < n / 4; i++) { sum ^= w[i]; sum *= 16777619u; }
            h->check_sum = sum; h->check_words = (uint32_t)(n / 4); h->checked = 1;
            if (h->outp) {
                char path[512];
                if (h->rank == 0) snprintf(path, sizeof path, "%s", h->outp); else snprintf(path, sizeof path, "%s.dev%d", h->outp, h->rank);
                FILE *f = fopen(path, "wb");
                if (!f) { free(w); SHARD_FAIL("-o %.150s", path); }
                fwrite(w, 1, n, f); fclose(f);
            }
            free(w);
        }
    } else if (S) {
        h->frames = (double)S * (double)frames * h->calls;      /* dry run: what this shard WOULD have processed */
    }
out:
    /* a shard that failed (or has no streams) still meets the others at the barriers */
    if (!h->dry) for (; in_barrier < 2; in_barrier++) pthread_barrier_wait(h->bar);
    if (d_pcm) (void)hipFree(d_pcm);
    if (d_pairs) (void)hipFree(d_pairs);
    if (d_sub) (void)hipFree(d_sub);
    if (d_peaks) (void)hipFree(d_peaks);
    if (ctx) dspi_destroy(ctx);
    return NULL;
#undef SHARD_FAIL
}

/* sum(frames), max(seconds) over the devices: ONE collective on 8-byte records — ncclAllReduce over the
 * node's communicator (single process: ncclCommInitAll + a group call); -D none: the same reduction on the host */
static int reduce_node(Shard *sh, int n, int dry, double *frames, double *seconds, const char **how) {
    if (dry) {
        *frames = 0.0; *seconds = 0.0;
        for (int i = 0; i < n; i++) { *frames += sh[i].frames; if (sh[i].seconds > *seconds) *seconds = sh[i].seconds; }
        *how = "host stand-in (no device)";
        return 0;
    }
#ifndef DSPI_HOST_RCCL
    *frames = 0.0; *seconds = 0.0;
    for (int i = 0; i < n; i++) { *frames += sh[i].frames; if (sh[i].seconds > *seconds) *seconds = sh[i].seconds; }
    *how = "host reduction (built without RCCL: no rccl/rccl.h in this ROCm install)";
    return 0;
#else
    ncclComm_t *comm = (ncclComm_t *)calloc((size_t)n, sizeof(ncclComm_t));
    hipStream_t *st = (hipStream_t *)calloc((size_t)n, sizeof(hipStream_t));
    double **d = (double **)calloc((size_t)n, sizeof(double *));
    int *devs = (int *)malloc(sizeof(int) * (size_t)n);
    int rc = 1;
    for (int i = 0; i < n; i++) devs[i] = i;
    if (ncclCommInitAll(comm, n, devs) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll(%d) failed\n", n); goto out; }
    for (int i = 0; i < n; i++) {
        const double v[2] = {sh[i].frames, sh[i].seconds};
        if (hipSetDevice(i) != hipSuccess || hipStreamCreate(&st[i]) != hipSuccess || hipMalloc((void **)&d[i], 32) != hipSuccess ||
            hipMemcpy(d[i], v, 16, hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "reduce: device %d\n", i); goto out; }
    }
    ncclGroupStart();
    for (int i = 0; i < n; i++) ncclAllReduce(d[i], d[i] + 2, 1, ncclDouble, ncclSum, comm[i], st[i]);
    for (int i = 0; i < n; i++) ncclAllReduce(d[i] + 1, d[i] + 3, 1, ncclDouble, ncclMax, comm[i], st[i]);
    if (ncclGroupEnd() != ncclSuccess) { fprintf(stderr, "ncclAllReduce failed\n"); goto out; }
    for (int i = 0; i < n; i++) { (void)hipSetDevice(i); if (hipStreamSynchronize(st[i]) != hipSuccess) goto out; }
    {
        double r[2];
        (void)hipSetDevice(0);
        if (hipMemcpy(r, d[0] + 2, 16, hipMemcpyDeviceToHost) != hipSuccess) goto out;
        *frames = r[0]; *seconds = r[1];
    }
    *how = "rccl: ncclAllReduce(sum frames) + ncclAllReduce(max seconds), 8 bytes each, one process, ncclCommInitAll";
    rc = 0;
out:
    for (int i = 0; i < n; i++) {
        (void)hipSetDevice(i);
        if (d[i]) (void)hipFree(d[i]);
        if (st[i]) (void)hipStreamDestroy(st[i]);
        if (comm[i]) ncclCommDestroy(comm[i]);
    }
    free(comm); free(st); free(d); free(devs);
    return rc;
#endif
}

static int node_run(int n_gpus, int strong, int dry, int flavor, uint32_t streams, uint32_t rate, uint32_t block_len, uint32_t blocks, uint32_t calls,
                    uint32_t warmup, double vol_db, const char *bulk, const char *slot, const char *outp) {
    if (n_gpus < 1 || n_gpus > 64 || !calls || !streams) { fprintf(stderr, "-g: 1..64 devices, -c >= 1, -s >= 1\n"); return 2; }
    if (!dry) {
        int have = 0;
        if (hipGetDeviceCount(&have) != hipSuccess || have < n_gpus) { fprintf(stderr, "-g %d: %d device(s) visible\n", n_gpus, have); return 1; }
    }
    size_t bulk_len = 0, slot_len = 0;
    void *bulk_b = bulk ? slurp(bulk, &bulk_len) : NULL, *slot_b = slot ? slurp(slot, &slot_len) : NULL;
    const uint32_t total = strong ? streams : streams * (uint32_t)n_gpus;
    Shard *sh = (Shard *)calloc((size_t)n_gpus, sizeof(Shard));
    pthread_t *th = (pthread_t *)calloc((size_t)n_gpus, sizeof(pthread_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)n_gpus);
    for (int i = 0; i < n_gpus; i++) {
        Shard *h = &sh[i];
        h->rank = i; h->world = n_gpus; h->dry = dry; h->flavor = flavor;
        stream_range((uint32_t)i, (uint32_t)n_gpus, total, &h->first, &h->last);
        h->rate = rate; h->block_len = block_len; h->blocks = blocks; h->calls = calls; h->warmup = warmup; h->vol_db = vol_db;
        h->bulk = bulk_b; h->bulk_len = bulk_len; h->slot = slot_b; h->slot_len = slot_len; h->outp = outp; h->bar = &bar;
        pthread_create(&th[i], NULL, shard_main, h);
    }
    int rc = 0, channels = 0;
    for (int i = 0; i < n_gpus; i++) {
        pthread_join(th[i], NULL);
        if (sh[i].rc) { fprintf(stderr, "device %d: %s\n", i, sh[i].err); rc = 1; }
        if (sh[i].channels) channels = sh[i].channels;
    }
    pthread_barrier_destroy(&bar);
    double frames = 0.0, seconds = 0.0; const char *how = "";
    if (!rc) rc = reduce_node(sh, n_gpus, dry, &frames, &seconds, &how);
    if (!rc) {
        const double fps = seconds > 0.0 ? frames / seconds : 0.0;
        printf("{\"metric\": \"audio samples/s (whole node)\", \"value\": %.6e, \"unit\": \"samples/s\", \"n_gpus\": %d, \"steps\": %u, \"warmup\": %u, "
               "\"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"%s\", \"vs_baseline\": null, \"dtype\": \"%s\", \"data\": \"synthetic\", "
               "\"config\": {\"workload\": \"dspi_host -g: %u streams in total, %u Hz, %u packets of %u frames per call\", \"streams_total\": %u, \"channels\": %d, "
               "\"frames\": %.0f, \"seconds\": %.6f, \"frames_per_s\": %.6e, \"realtime_streams\": %.1f, \"parallelism\": \"one process, one context + feeder thread per GPU\", \"shards\": [",
               fps * channels, n_gpus, calls, warmup, seconds / calls * 1e3, strong ? "strong" : "weak", (flavor & 0xff) ? "f32" : "int32 (Q28)",
               total, rate, blocks, block_len, total, channels, frames, seconds, fps, fps / rate);
        for (int i = 0; i < n_gpus; i++) printf("%s[%u, %u]", i ? ", " : "", sh[i].first, sh[i].last);
        printf("], \"ms_per_step_per_device\": [");
        for (int i = 0; i < n_gpus; i++) printf("%s%.6f", i ? ", " : "", sh[i].seconds / calls * 1e3);
        printf("]}, ");
        {   /* roofline of the node: algorithmic bytes of every device's shard over the slowest device's time, against N x 8 TB/s */
            double strict_b = 0.0, resident_b = 0.0;
            for (int i = 0; i < n_gpus; i++) { const double f = sh[i].frames; strict_b += f * sh[i].bytes_strict; resident_b += f * sh[i].bytes_resident; }
            const double peak = 8000.0 * n_gpus, ach = seconds > 0.0 ? strict_b / seconds / 1e9 : 0.0, ach_r = seconds > 0.0 ? resident_b / seconds / 1e9 : 0.0;
            printf("\"roofline\": {\"bound\": \"hbm\", \"achieved\": %.3f, \"peak\": %.1f, \"unit\": \"GB/s\", \"frac\": %.6f, \"traffic\": null, "
                   "\"algorithmic_bytes_per_frame\": %.3f, \"algorithmic_bytes_per_frame_hbm_resident\": %.3f, \"frac_hbm_resident\": %.6f, "
                   "\"what\": \"strict bytes (bench.py:algorithmic_bytes) of all shards / the slowest device's wall time of %u calls; peak = n_gpus x 8 TB/s\"}, ",
                   ach, peak, ach / peak, frames > 0.0 ? strict_b / frames : 0.0, frames > 0.0 ? resident_b / frames : 0.0, ach_r / peak, calls);
        }
        printf("\"affinity\": [");
        for (int i = 0; i < n_gpus; i++) printf("%s{\"device\": %d, \"numa_node\": %d, \"cpus\": %d, \"source\": \"%s\"}", i ? ", " : "", i, sh[i].numa_node, sh[i].cpus_pinned, sh[i].pin_src);
        printf("], \"checked_streams\": [");
        for (int i = 0, k = 0; i < n_gpus; i++) if (sh[i].checked) printf("%s{\"device\": %d, \"stream\": %u, \"pair_words\": %u, \"fnv1a\": %u}", k++ ? ", " : "", i, sh[i].first, sh[i].check_words, sh[i].check_sum);
        printf("], \"dist\": {\"backend\": \"%s\", \"world_size\": %d}, \"dry_run\": %s}\n", how, n_gpus, dry ? "true" : "false");
    }
    free(sh); free(th); free(bulk_b); free(slot_b);
    return rc;
}

int main(int argc, char **argv) {
    int flavor = DSPI_FLAVOR_RP2350_F32;
    uint32_t streams = 4096, rate = 48000, block_len = 48, blocks = 100, calls = 10;
    const char *bulk = NULL, *slot = NULL, *in = NULL, *outp = NULL, *allp = NULL, *latp = NULL;
    double vol_db = 0.0;
    int rt = 0, n_gpus = 0, strong = 0, dry = 0;
    uint32_t warmup = 2;
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
        else if (!strcmp(a, "-g") && v) { n_gpus = atoi(v); i++; }
        else if (!strcmp(a, "-S") && v) { strong = !strcmp(v, "strong"); i++; }
        else if (!strcmp(a, "-w") && v) { warmup = (uint32_t)atoi(v); i++; }
        else if (!strcmp(a, "-D") && v) { dry = !strcmp(v, "none"); i++; }
        else if (!strcmp(a, "-O") && v) { allp = v; i++; }
        else if (!strcmp(a, "-L") && v) { latp = v; i++; }
        else { fprintf(stderr, "usage: %s [-f q28|f32|f32fma] [-s streams] [-r rate] [-b block_len] [-n blocks] [-c calls] [-B bulk.bin|-P slot.bin] [-i pcm.raw] [-o pairs.raw] [-v vol_db] [-rt] [-g gpus [-S weak|strong] [-w warmup] [-D none]]\n", argv[0]); return 2; }
    }
    if (n_gpus > 0) return node_run(n_gpus, strong, dry, flavor, streams, rate, block_len, blocks, calls, warmup, vol_db, bulk, slot, outp);
    if (dry) { fprintf(stderr, "-D none goes with -g\n"); return 2; }
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
    memset(&out, 0, sizeof out);
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
    if (outp) {
        FILE *f = fopen(outp, "wb");
        if (!f) { perror(outp); return 2; }
        fwrite(out.pairs, 8, (size_t)pairs_n * frames, f); fclose(f);
    }
    dspi_destroy(ctx);
    return 0;
}
