// Training-time augmentation on the device: horizontal flip -> bilinear resize -> crop -> float,
// i.e. the per-sample transform chain of the reference's OpenImages dataset (src/helpers/datasets.py:206-216:
// RandomHorizontalFlip, Resize((ceil(s H), ceil(s W))), RandomCrop(crop), ToTensor [, Normalize(.5,.5)]) for a whole
// batch of decoded uint8 images in one launch.  The reference runs this in PIL on DataLoader workers, one image at a
// time ("TODO: This definitely needs to be optimized", datasets.py:152).
//
// Integer-exact with Pillow's 8-bit resampler (libImaging/Resample.c): per axis fixed-point weights (22 fractional
// bits) over a clipped window, horizontal pass to an 8-bit intermediate, vertical pass, each clip8((sum + 2^21) >> 22).
// The weights of the crop window's rows/columns are computed on the host in double precision exactly as Pillow's
// precompute_coeffs does (hific_amd/helpers/augment.py) - ~10 KB per image - and only the crop's pixels are ever
// computed: each output pixel gathers its (<= kmax x kmax) source window straight from the uint8 HWC image.
// HBM-bound byte work: ~0.3 MB read + 0.8 MB written per 256x256 crop; one thread per output pixel, 3 channels.
#include "common.h"

#define AUG_PRECISION_BITS 22

struct HificAugImage {       // mirrors `hific_aug_image` in include/hific_hip.h
    const unsigned char* src;   // uint8 [H][W][3] (device)
    int H, W;                   // source size
    int flip;                   // 1: horizontal flip applied BEFORE the resize (source column W-1-x)
    int resize_x, resize_y;     // 0: that axis keeps its size (Pillow skips the pass: no rounding through weights)
    int top, left;              // crop origin in the resized image (only used when the axis is not resized)
};

__device__ __forceinline__ int aug_clip8(int v) {
    v >>= AUG_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// xb/yb: [B][crop][2] = (first source index, tap count) of each crop column / row; xk/yk: [B][crop][kmax] weights
__global__ __launch_bounds__(256) void augment_crop_kernel(const HificAugImage* __restrict__ imgs,
                                                           const int* __restrict__ xb, const int* __restrict__ xk,
                                                           const int* __restrict__ yb, const int* __restrict__ yk,
                                                           int crop, int kmax, int normalize, float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= crop * crop) return;
    const int oy = p / crop, ox = p - oy * crop;
    const HificAugImage im = imgs[b];
    const unsigned char* src = im.src;
    const int half = 1 << (AUG_PRECISION_BITS - 1);
    int x0, xn, y0, yn;
    const int* kx = xk + ((size_t)b * crop + ox) * kmax;
    const int* ky = yk + ((size_t)b * crop + oy) * kmax;
    if (im.resize_x) { x0 = xb[((size_t)b * crop + ox) * 2]; xn = xb[((size_t)b * crop + ox) * 2 + 1]; }
    else { x0 = im.left + ox; xn = 1; }
    if (im.resize_y) { y0 = yb[((size_t)b * crop + oy) * 2]; yn = yb[((size_t)b * crop + oy) * 2 + 1]; }
    else { y0 = im.top + oy; yn = 1; }
    int v0 = half, v1 = half, v2 = half;
    int h0 = 0, h1 = 0, h2 = 0;
    for (int yi = 0; yi < yn; ++yi) {
        const unsigned char* row = src + (size_t)(y0 + yi) * im.W * 3;
        if (im.resize_x) {
            int a0 = half, a1 = half, a2 = half;
            for (int xi = 0; xi < xn; ++xi) {
                int sx = x0 + xi;
                if (im.flip) sx = im.W - 1 - sx;
                const int w = kx[xi];
                const unsigned char* px = row + sx * 3;
                a0 += (int)px[0] * w; a1 += (int)px[1] * w; a2 += (int)px[2] * w;
            }
            h0 = aug_clip8(a0); h1 = aug_clip8(a1); h2 = aug_clip8(a2);
        } else {
            const int sx = im.flip ? im.W - 1 - x0 : x0;
            const unsigned char* px = row + sx * 3;
            h0 = px[0]; h1 = px[1]; h2 = px[2];
        }
        if (im.resize_y) { const int w = ky[yi]; v0 += h0 * w; v1 += h1 * w; v2 += h2 * w; }
    }
    int r0, r1, r2;
    if (im.resize_y) { r0 = aug_clip8(v0); r1 = aug_clip8(v1); r2 = aug_clip8(v2); }
    else { r0 = h0; r1 = h1; r2 = h2; }
    // ToTensor: uint8 / 255 (float32 division, as torch does); optional Normalize((.5,.5,.5),(.5,.5,.5))
    float f0 = (float)r0 / 255.f, f1 = (float)r1 / 255.f, f2 = (float)r2 / 255.f;
    if (normalize) { f0 = (f0 - 0.5f) / 0.5f; f1 = (f1 - 0.5f) / 0.5f; f2 = (f2 - 0.5f) / 0.5f; }
    const size_t plane = (size_t)crop * crop;
    float* o = out + (size_t)b * 3 * plane + p;
    o[0] = f0; o[plane] = f1; o[2 * plane] = f2;
}

extern "C" int hific_augment_crop(const void* imgs, const int* xb, const int* xk, const int* yb, const int* yk, int B,
                                  int crop, int kmax, int normalize, float* out, hipStream_t st) {
    if (!imgs || !xb || !xk || !yb || !yk || !out || B <= 0 || crop <= 0 || kmax <= 0) return HIFIC_ERR_ARG;
    dim3 grid(cdiv(crop * crop, 256), B);
    hipLaunchKernelGGL(augment_crop_kernel, grid, dim3(256), 0, st, (const HificAugImage*)imgs, xb, xk, yb, yk, crop,
                       kmax, normalize, out);
    return hific_launch_status();
}
