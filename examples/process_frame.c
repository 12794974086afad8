/* examples/process_frame.c -- the C ABI from plain C: one frame through the trait sequence
 * (load_frame -> crop_to_map -> isolate -> mask -> find_marker_lines), the call order of
 * src/vision/mod.rs:36-240.  Input: a raw BGRA8 file (w * h * 4 bytes).
 *
 *   gcc -std=c99 -Iinclude examples/process_frame.c -Lsquad-mortar-helper_amd -lsmh_vision_hip \
 *       -Wl,-rpath,$PWD/squad-mortar-helper_amd -o process_frame
 *   ./process_frame frame.bgra 2560 1440
 */
#include <stdio.h>
#include <stdlib.h>

#include "smh_vision_hip.h"

static void log_sink(int level, const char *msg) { fprintf(stderr, "[smh %d] %s\n", level, msg); }

#define CHECK(call) do { int rc_ = (call); if (rc_ != SMHV_OK) { fprintf(stderr, "%s: error %d: %s\n", #call, rc_, smhv_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
	if (argc != 4) { fprintf(stderr, "usage: %s frame.bgra width height\n", argv[0]); return 2; }
	const uint32_t w = (uint32_t)atoi(argv[2]), h = (uint32_t)atoi(argv[3]);
	const size_t bytes = (size_t)w * h * 4;
	uint8_t *frame = (uint8_t *)malloc(bytes);
	FILE *f = fopen(argv[1], "rb");
	if (!frame || !f || fread(frame, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, argv[1]); return 2; }
	fclose(f);

	smhv_ctx *ctx = NULL;
	CHECK(smhv_init(0, log_sink, &ctx));                 /* fails with SMHV_E_NO_DEVICE without a gfx950 GPU: the caller falls back */
	uint32_t roi[4];
	CHECK(smhv_map_bounds(w, h, roi));
	uint8_t *ui = (uint8_t *)malloc((size_t)roi[2] * roi[3] * 4);
	int map_open = 0;
	CHECK(smhv_load_frame(ctx, frame, w, h));
	CHECK(smhv_crop_to_map(ctx, 1, &map_open, roi, ui));
	if (!map_open) { printf("map closed\n"); smhv_shutdown(ctx); return 0; }
	CHECK(smhv_isolate_map_markers(ctx));
	CHECK(smhv_mask_marker_lines(ctx));
	smhv_line lines[SMHV_MAX_LINES];
	uint32_t n = 0;
	CHECK(smhv_find_marker_lines(ctx, 15, lines, &n));  /* max_gap 15: src/vision/mod.rs:112-115 */
	printf("map ROI %u,%u %ux%u, %u marker line(s)\n", roi[0], roi[1], roi[2], roi[3], n);
	for (uint32_t i = 0; i < n; ++i) printf("  (%.1f, %.1f) -> (%.1f, %.1f)\n", lines[i].x0, lines[i].y0, lines[i].x1, lines[i].y1);
	smhv_shutdown(ctx);
	free(ui); free(frame);
	return 0;
}
