// pose_math.h -- host-side 4x4 helpers of the TSDF path, in the reference's exact fp32
// operation order so that the relative pose handed to the kernel is bit-identical to the
// one TSDF::Integrate builds (ref: src/tsdf.cu:142, :253-273, :276-403).
//
// The inverse is the reference's cofactor expansion, written here as a table: cofactor k is
// the signed sum, left to right, of six triple products m[a]*m[b]*m[c] (each product itself
// left to right).  Negating a product or its first factor is exact in IEEE arithmetic, so
// evaluating "acc += sign * (a*b*c)" rounds exactly as the reference's expression does.
#pragma once
#include <cstdint>
#include <cstring>

namespace tsdf_host {

inline void multiply_matrix(const float *a, const float *b, float *out)
{
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = a[4 * i] * b[j];
            for (int k = 1; k < 4; ++k) acc = acc + a[4 * i + k] * b[4 * k + j];
            r[4 * i + j] = acc;
        }
    std::memcpy(out, r, sizeof r);
}

struct Term { int8_t sign, a, b, c; };

// clang-format off
static const Term kCofactor[16][6] = {
 /* 0*/ {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}},
 /* 1*/ {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}},
 /* 2*/ {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}},
 /* 3*/ {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}},
 /* 4*/ {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}},
 /* 5*/ {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}},
 /* 6*/ {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}},
 /* 7*/ {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}},
 /* 8*/ {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}},
 /* 9*/ {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}},
 /*10*/ {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}},
 /*11*/ {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}},
 /*12*/ {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}},
 /*13*/ {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}},
 /*14*/ {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}},
 /*15*/ {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}},
};
// clang-format on

// Returns false (and leaves inv_out untouched) when the determinant is exactly zero,
// as ref: src/tsdf.cu:394-395.
inline bool invert_matrix(const float *m, float *inv_out)
{
    float cof[16];
    for (int k = 0; k < 16; ++k) {
        float acc = 0.0f;
        for (int t = 0; t < 6; ++t) {
            const Term &q = kCofactor[k][t];
            float prod = m[q.a] * m[q.b] * m[q.c];
            if (t == 0) acc = q.sign > 0 ? prod : -prod;
            else acc = q.sign > 0 ? acc + prod : acc - prod;
        }
        cof[k] = acc;
    }
    float det = m[0] * cof[0] + m[1] * cof[4] + m[2] * cof[8] + m[3] * cof[12];
    if (det == 0) return false;
    det = (float)(1.0 / (double)det);  // ref: src/tsdf.cu:397 (double reciprocal, stored to float)
    for (int i = 0; i < 16; ++i) inv_out[i] = cof[i] * det;
    return true;
}

}  // namespace tsdf_host
