/*
 * Test infrastructure: the pl_gpu API contract from plain C, after the reference's
 * pl_buffer_tests / pl_texture_tests (src/tests/gpu_tests.c:27-130, 195-330): buffer and
 * texture round trips, and -- the point of this file -- that every misuse the reference's
 * validation front-end rejects (src/gpu.c:440-497, 543-716) is rejected here too instead of
 * becoming an out-of-bounds device access. Runs on the GPU box (tests/test_gpu_c_abi.py).
 * Exit status 0 and a final "ok" line = every check held.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>
#include <libplacebo/utils/upload.h>

static int errors_logged;

static void count_errors(void *priv, enum pl_log_level level, const char *msg)
{
    (void) priv; (void) msg;
    if (level <= PL_LOG_ERR)
        errors_logged++;
}

#define CHECK(x) do { if (!(x)) { printf("FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)
// the call must fail AND say why
#define REJECTED(call) do { const int before = errors_logged; call; \
    if (errors_logged == before) { printf("FAILED line %d: no error logged by %s\n", __LINE__, #call); return 1; } } while (0)

int main(void)
{
    pl_log log = pl_log_create(PL_API_VER, pl_log_params(.log_cb = count_errors,
                                                         .log_level = PL_LOG_ERR));
    pl_hip hip = pl_hip_create(log, pl_hip_params(.device = 0));
    CHECK(hip);
    pl_gpu gpu = hip->gpu;

    /* ---- buffers ------------------------------------------------------------------------- */
    uint8_t pattern[256], back[256];
    for (int i = 0; i < 256; i++)
        pattern[i] = (uint8_t) (i * 7 + 3);
    pl_buf rw = pl_buf_create(gpu, pl_buf_params(.size = 256, .host_readable = true,
                                                 .host_writable = true));
    pl_buf dev = pl_buf_create(gpu, pl_buf_params(.size = 128, .storable = true));
    CHECK(rw && dev);
    pl_buf_write(gpu, rw, 0, pattern, 256);
    CHECK(pl_buf_read(gpu, rw, 0, back, 256) && !memcmp(back, pattern, 256));
    pl_buf_write(gpu, rw, 64, pattern, 32);
    CHECK(pl_buf_read(gpu, rw, 60, back, 40));
    CHECK(!memcmp(back, pattern + 60, 4) && !memcmp(back + 4, pattern, 32) && !memcmp(back + 36, pattern + 96, 4));
    CHECK(errors_logged == 0);

    memset(back, 0xaa, sizeof(back));
    REJECTED(pl_buf_write(gpu, rw, 200, pattern, 100));            // past the end
    REJECTED(pl_buf_write(gpu, rw, (size_t) -8, pattern, 16));     // offset + size wraps
    REJECTED(pl_buf_write(gpu, rw, 2, pattern, 4));                // unaligned offset
    REJECTED(pl_buf_write(gpu, dev, 0, pattern, 4));               // not host_writable
    REJECTED(CHECK(!pl_buf_read(gpu, rw, 250, back, 7)));
    REJECTED(CHECK(!pl_buf_read(gpu, dev, 0, back, 4)));           // not host_readable
    CHECK(back[0] == 0xaa);
    REJECTED(pl_buf_copy(gpu, dev, 100, rw, 0, 64));               // dst range
    REJECTED(pl_buf_copy(gpu, dev, 0, rw, 224, 64));               // src range
    REJECTED(pl_buf_copy(gpu, rw, 0, rw, 128, 64));                // same buffer
    REJECTED(CHECK(!pl_buf_create(gpu, pl_buf_params(.size = 0))));
    REJECTED(CHECK(!pl_buf_create(gpu, pl_buf_params(.size = 64, .export_handle = PL_HANDLE_FD))));
    // no host-visible mappings on this backend (limits.max_mapped_size == 0): refused, never a
    // buffer whose `data` is NULL (src/gpu.c:595-611)
    REJECTED(CHECK(!pl_buf_create(gpu, pl_buf_params(.size = 64, .host_mapped = true))));
    {
        // pl_buf_recreate keeps a buffer only if it covers every capability asked for (:630-641)
        pl_buf b = pl_buf_create(gpu, pl_buf_params(.size = 128, .host_writable = true));
        CHECK(b);
        pl_buf first = b;
        CHECK(pl_buf_recreate(gpu, &b, pl_buf_params(.size = 96, .host_writable = true)) && b == first);
        CHECK(pl_buf_recreate(gpu, &b, pl_buf_params(.size = 96, .host_writable = true, .host_readable = true)));
        CHECK(b->params.host_readable && b->params.size == 96);
        uint8_t probe[4] = {0};
        CHECK(pl_buf_read(gpu, b, 0, probe, 4));
        // ... and never takes initial contents: an error, the buffer stays what it was (:647-650)
        pl_buf kept = b;
        REJECTED(CHECK(!pl_buf_recreate(gpu, &b, pl_buf_params(.size = 96, .host_writable = true,
                                                               .host_readable = true, .initial_data = probe))));
        CHECK(b == kept);
        pl_buf_destroy(gpu, &b);
    }
    // none of the above touched the buffer
    CHECK(pl_buf_read(gpu, rw, 0, back, 256));
    CHECK(!memcmp(back, pattern, 64) && !memcmp(back + 64, pattern, 32) && !memcmp(back + 96, pattern + 96, 160));
    int before = errors_logged;
    pl_buf_copy(gpu, dev, 16, rw, 0, 64);
    pl_buf_copy(gpu, rw, 128, dev, 16, 64);
    CHECK(pl_buf_read(gpu, rw, 128, back, 64) && !memcmp(back, pattern, 64));
    CHECK(errors_logged == before);

    /* ---- textures ------------------------------------------------------------------------ */
    pl_fmt fmt = pl_find_named_fmt(gpu, "rgba8");
    CHECK(fmt);
    enum { W = 16, H = 8 };
    uint8_t img[H][W][4], out[H][W][4];
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < 4; c++)
                img[y][x][c] = (uint8_t) (y * 31 + x * 5 + c);
    pl_tex tex = pl_tex_create(gpu, pl_tex_params(.w = W, .h = H, .format = fmt, .sampleable = true,
                                                  .host_writable = true, .host_readable = true));
    pl_tex ro = pl_tex_create(gpu, pl_tex_params(.w = W, .h = H, .format = fmt, .sampleable = true));
    CHECK(tex && ro);
    before = errors_logged;
    CHECK(pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .ptr = img)));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex, .ptr = out)));
    CHECK(!memcmp(img, out, sizeof(img)));
    // through a buffer, a sub-rectangle with a padded pitch
    pl_buf stage = pl_buf_create(gpu, pl_buf_params(.size = 4 * 80, .host_readable = true,
                                                    .host_writable = true));
    CHECK(stage);
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex, .buf = stage, .buf_offset = 16,
                          .row_pitch = 80, .rc = { .x0 = 2, .y0 = 1, .x1 = 12, .y1 = 4 })));
    uint8_t rows[4 * 80];
    CHECK(pl_buf_read(gpu, stage, 0, rows, sizeof(rows)));
    for (int y = 0; y < 3; y++)
        CHECK(!memcmp(rows + 16 + 80 * y, &img[1 + y][2][0], 40));
    CHECK(errors_logged == before);

    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = ro, .ptr = img))));          // not host_writable
    REJECTED(CHECK(!pl_tex_download(gpu, pl_tex_transfer_params(.tex = ro, .ptr = out))));
    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex))));                     // neither ptr nor buf
    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .ptr = img, .buf = stage))));  // both
    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .ptr = img,
                                  .rc = { .x0 = 4, .x1 = W + 1, .y1 = H }))));                    // rect outside
    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .ptr = img, .row_pitch = 32))));   // pitch < row
    {   // a pitch that is not a multiple of the texel alignment (2 bytes for 16-bit samples)
        static uint16_t img16[H + 1][W][4];
        pl_tex t16 = pl_tex_create(gpu, pl_tex_params(.w = W, .h = H, .format = pl_find_named_fmt(gpu, "rgba16"),
                                                      .sampleable = true, .host_writable = true));
        CHECK(t16);
        REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = t16, .ptr = img16, .row_pitch = W * 8 + 1))));
        CHECK(pl_tex_upload(gpu, pl_tex_transfer_params(.tex = t16, .ptr = img16, .row_pitch = W * 8 + 2)));
        pl_tex_destroy(gpu, &t16);
    }
    // 8 rows of 64 bytes do not fit 320 bytes at offset 0; nor 4 rows at offset 80
    REJECTED(CHECK(!pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex, .buf = stage))));
    REJECTED(CHECK(!pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex, .buf = stage, .buf_offset = 80,
                                    .rc = { .x1 = W, .y1 = 4 }))));
    REJECTED(CHECK(!pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .buf = stage,
                                  .buf_offset = (size_t) -64, .rc = { .x1 = W, .y1 = 1 }))));
    // the stage buffer still holds exactly what the valid download left
    uint8_t rows2[4 * 80];
    CHECK(pl_buf_read(gpu, stage, 0, rows2, sizeof(rows2)) && !memcmp(rows, rows2, sizeof(rows)));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex, .ptr = out)) && !memcmp(img, out, sizeof(img)));

    /* ---- 1D plane with byte-swapped samples (height 0) --------------------------------------- */
    uint16_t line[32], swapped[32], got[32];
    for (int i = 0; i < 32; i++) {
        line[i] = (uint16_t) (i * 2053 + 77);
        swapped[i] = (uint16_t) (line[i] << 8 | line[i] >> 8);
    }
    pl_tex tex1d = NULL;
    struct pl_plane plane;
    before = errors_logged;
    CHECK(pl_upload_plane(gpu, &plane, &tex1d, &(struct pl_plane_data) {
        .type = PL_FMT_UNORM, .width = 32, .height = 0, .component_size = {16},
        .component_map = {0}, .pixel_stride = 2, .pixels = swapped, .swapped = true,
    }));
    CHECK(tex1d && tex1d->params.w == 32 && tex1d->params.h == 0);
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = tex1d, .ptr = got)));
    CHECK(!memcmp(got, line, sizeof(line)));
    CHECK(errors_logged == before);

    /* ---- pl_pass is a stated deviation: it links, and fails loudly ----------------------------- */
    REJECTED(CHECK(!pl_pass_create(gpu, pl_pass_params(.type = PL_PASS_COMPUTE, .glsl_shader = "void main() {}"))));

    pl_tex_destroy(gpu, &tex1d);
    pl_tex_destroy(gpu, &tex);
    pl_tex_destroy(gpu, &ro);
    pl_buf_destroy(gpu, &stage);
    pl_buf_destroy(gpu, &rw);
    pl_buf_destroy(gpu, &dev);
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    puts("ok");
    return 0;
}
