/* ORACLE (test infrastructure, never shipped, never linked into the product).
 *
 * CPU restatement of cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) as the
 * reference calls it at src/functions/inference.py:1164 and :2605.  The arithmetic lives in the
 * un-vendored dependency opencv-python-headless == 4.11.0.86 (requirements.txt:31), absent from
 * /root/reference and not installable here, so this file restates the published algorithm
 * (Suzuki-Abe border following as implemented by OpenCV's cvFindNextContour / icvFetchContour:
 * zero-padded 0/1 image, raster scan, outer border starts at a 0 -> 1 transition whose last
 * labelled neighbour on the row is not an open (positive-valued) border, 8-neighbourhood search
 * starting from the west, "right bound" pixels labelled nbd|-128, SIMPLE keeps a point whenever the
 * outgoing chain direction changes).  PARITY UNPINNED: the reference holds no contour fixtures.
 *
 * Contours are returned in DISCOVERY (raster) order; OpenCV's Python binding returns the reverse.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* returns number of contours, or -1 if a capacity was exceeded */
int demia_ref_find_external_contours(const uint8_t* mask, int H, int W, int32_t* pts, int max_pts,
                                     int32_t* offsets /* [max_contours + 1] */, int max_contours) {
    const int step = W + 2;
    signed char* img = (signed char*)calloc((size_t)(H + 2) * step, 1);
    if (!img) return -1;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) img[(y + 1) * step + x + 1] = mask[(size_t)y * W + x] ? 1 : 0;
    int deltas[16];
    for (int i = 0; i < 8; ++i) deltas[i] = deltas[i + 8] = DY[i] * step + DX[i];
    const signed char nbd = 2;
    int ncont = 0, npts = 0;
    offsets[0] = 0;
    for (int y = 1; y <= H; ++y) {
        signed char* row = img + (size_t)y * step;
        int prev = 0;
        int lnbd_x = 0; /* lnbd.y == y */
        for (int x = 1; x <= W + 1; ++x) {
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) goto resume_scan;
                if (prev & -2) lnbd_x = x - 1;
                is_hole = 1;
            }
            if (is_hole || row[lnbd_x] > 0) goto resume_scan; /* RETR_EXTERNAL */
            {
                lnbd_x = x;
                /* ---- icvFetchContour(row + x, step, (x-1, y-1), CHAIN_APPROX_SIMPLE) ---- */
                signed char* i0 = row + x;
                signed char *i1, *i3, *i4 = 0;
                int s, s_end, prev_s;
                int px = x - 1, py = y - 1; /* un-padded coordinates */
                if (ncont >= max_contours) { free(img); return -1; }
                s_end = s = 4;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) { /* single pixel */
                    *i0 = (signed char)(nbd | -128);
                    if (npts + 1 > max_pts) { free(img); return -1; }
                    pts[2 * npts] = px; pts[2 * npts + 1] = py; ++npts;
                } else {
                    i3 = i0;
                    prev_s = s ^ 4;
                    for (;;) {
                        s_end = s;
                        for (;;) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
                        else if (*i3 == 1) *i3 = nbd;
                        if (s != prev_s) {
                            if (npts + 1 > max_pts) { free(img); return -1; }
                            pts[2 * npts] = px; pts[2 * npts + 1] = py; ++npts;
                            prev_s = s;
                        }
                        px += DX[s]; py += DY[s];
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                ++ncont;
                offsets[ncont] = npts;
                p = row[x];
            }
        resume_scan:
            prev = p;
            if (prev & -2) lnbd_x = x;
        }
    }
    free(img);
    return ncont;
}
