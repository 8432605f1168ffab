/* Plain-C client of the detector groups of include/aprilgrid_amd.h: what a Rust host does to drive
 * every GPU of a node from ONE process without torch -- one detector per device, the batch sharded
 * by frame (frames are independent: reference detect(&self), src/detector.rs:505), every rank's
 * saddle chain on its own device and stream, the per-rank result slabs gathered to device 0 (RCCL
 * send/recv over xGMI, or peer copies), one fetch for all frames.  Build:
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_group_client.c \
 *       -Laprilgrid-rs_amd -laprilgrid_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/aprilgrid-rs_amd -o c_group_client
 * Usage: c_group_client <raw L8 file with n_ranks*frames_per_rank frames> <width> <height> <frames_per_rank> <n_ranks> <rccl|peer>
 *        (n_ranks ranks are placed round-robin on the visible devices; "rccl" needs n_ranks <= devices)
 * Prints one line per frame: "frame <global index>: <saddles>" and a total. */
#include "aprilgrid_amd.h"

#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    if (argc != 7) {
        fprintf(stderr, "usage: %s <raw L8 frames> <width> <height> <frames_per_rank> <n_ranks> <rccl|peer>\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[2]), h = atoi(argv[3]), fpr = atoi(argv[4]), n = atoi(argv[5]);
    const int transport = strcmp(argv[6], "rccl") == 0 ? AGX_GATHER_RCCL : AGX_GATHER_PEER;
    const size_t frame_bytes = (size_t)w * h, shard_bytes = frame_bytes * (size_t)fpr;
    unsigned char *host = (unsigned char *)malloc(shard_bytes * (size_t)n);
    FILE *f = fopen(argv[1], "rb");
    if (!host || !f || fread(host, 1, shard_bytes * (size_t)n, f) != shard_bytes * (size_t)n) {
        fprintf(stderr, "cannot read %d frames of %dx%d from %s\n", fpr * n, w, h, argv[1]);
        return 2;
    }
    fclose(f);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
        fprintf(stderr, "no HIP device\n");
        return 1;
    }
    int devices[64];
    const void *d_frames[64];
    if (n < 1 || n > 64) return 2;
    for (int r = 0; r < n; ++r) {  /* rank r's shard goes to ITS device */
        void *p = NULL;
        devices[r] = r % n_dev;
        if (hipSetDevice(devices[r]) != hipSuccess || hipMalloc(&p, shard_bytes) != hipSuccess ||
            hipMemcpy(p, host + (size_t)r * shard_bytes, shard_bytes, hipMemcpyHostToDevice) != hipSuccess) {
            fprintf(stderr, "upload to device %d failed\n", devices[r]);
            return 1;
        }
        d_frames[r] = p;
    }
    int family = 0;
    agx_family_from_str("t36h11", &family);
    agx_group *grp = NULL;
    int st = agx_group_create(family, NULL, devices, n, transport, &grp);
    if (st != AGX_OK) {
        fprintf(stderr, "agx_group_create: %s (%s)\n", agx_status_string(st), agx_group_last_error(NULL));
        return 1;
    }
    const uint32_t cap = 2048;
    agx_saddle *out = (agx_saddle *)malloc(sizeof(agx_saddle) * cap * (size_t)(n * fpr));
    uint32_t *counts = (uint32_t *)calloc((size_t)(n * fpr), sizeof(uint32_t));
    for (int rep = 0; rep < 2; ++rep) {  /* twice: the slabs and communicators are reused */
        st = agx_group_saddles_enqueue(grp, d_frames, fpr, w, h, (size_t)w, frame_bytes, AGX_L8, 0);
        if (st == AGX_OK) st = agx_group_saddles_fetch(grp, out, cap, counts, NULL);
        if (st != AGX_OK) {
            fprintf(stderr, "group batch: %s (%s)\n", agx_status_string(st), agx_group_last_error(grp));
            return 1;
        }
    }
    unsigned long total = 0;
    for (int i = 0; i < n * fpr; ++i) {
        printf("frame %d: %u\n", i, counts[i]);
        total += counts[i];
    }
    printf("ranks %d on %d device(s), %s gather: %lu saddles in %d frames\n", agx_group_size(grp), n_dev,
           transport == AGX_GATHER_RCCL ? "rccl" : "peer", total, n * fpr);
    agx_group_destroy(grp);
    for (int r = 0; r < n; ++r) {
        hipSetDevice(devices[r]);
        hipFree((void *)d_frames[r]);
    }
    free(out);
    free(counts);
    free(host);
    return 0;
}
