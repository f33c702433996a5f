// NeRF backbone builder: 8 internal angles per residue -> N, CA, C, O coordinates, batched over
// pockets (one thread per pocket: the chain is a sequential dependency of 3 atom placements per
// residue; a 256-pocket batch is one workgroup).  Follows the reference's numpy path
// (structure_model/create_pdb.py:104-155,175-234): float32 trigonometry on the float32 angles, the
// frame algebra in float64 from the float64 initial coordinates.
#include "e3d_common.h"

namespace {

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 unit(V3 a) {
    const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / n, a.y / n, a.z / n};
}

// d with |cd| = len, angle(b,c,d) = bond_angle, dihedral(a,b,c,d) = torsion (create_pdb.py:175-234)
__device__ __forceinline__ V3 place_dihedral(V3 a, V3 b, V3 c, float bond_angle, float len, float torsion) {
    const V3 ab = sub(b, a);
    const V3 bc = unit(sub(c, b));
    const V3 n = unit(cross(ab, bc));
    const V3 nbc = cross(n, bc);
    const double d0 = (double)(-len * cosf(bond_angle));
    const double d1 = (double)(len * cosf(torsion) * sinf(bond_angle));
    const double d2 = (double)(len * sinf(torsion) * sinf(bond_angle));
    return {bc.x * d0 + nbc.x * d1 + n.x * d2 + c.x, bc.y * d0 + nbc.y * d1 + n.y * d2 + c.y,
            bc.z * d0 + nbc.z * d1 + n.z * d2 + c.z};
}

__device__ __forceinline__ void put(double* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

__global__ __launch_bounds__(256) void nerf_backbone_kernel(const float* __restrict__ angles,
                                                            const int32_t* __restrict__ lengths,
                                                            double* __restrict__ coords, int center, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int len = min(max(lengths[b], 0), L);
    const float* ang = angles + (int64_t)b * L * 8;   // phi psi omega dihedral_o tau CA:C:1N 1C:N:CA CA:C:O
    double* out = coords + (int64_t)b * L * 12;
    V3 n = {17.047, 14.099, 3.625}, ca = {16.967, 12.784, 4.338}, c = {15.685, 12.755, 5.133};
    double sx = 0, sy = 0, sz = 0;
    for (int i = 0; i < len; ++i) {
        const float* a = ang + i * 8;
        if (i > 0) {   // residue i from residue i-1: psi/omega of i-1, phi of i; bond angles of i-1
            const float* p = a - 8;
            const V3 n2 = place_dihedral(n, ca, c, p[5], 1.34f, p[1]);      // C-N : CA:C:1N, psi
            const V3 ca2 = place_dihedral(ca, c, n2, p[6], 1.46f, p[2]);    // N-CA: 1C:N:CA, omega
            const V3 c2 = place_dihedral(c, n2, ca2, p[4], 1.54f, a[0]);    // CA-C: tau, phi
            n = n2; ca = ca2; c = c2;
        }
        const V3 o = place_dihedral(n, ca, c, a[7], 1.22f, a[3]);          // C-O : CA:C:O, dihedral_o
        put(out + i * 12, n); put(out + i * 12 + 3, ca); put(out + i * 12 + 6, c); put(out + i * 12 + 9, o);
        sx += n.x + ca.x + c.x + o.x; sy += n.y + ca.y + c.y + o.y; sz += n.z + ca.z + c.z + o.z;
    }
    if (center && len > 0) {
        const double inv = 1.0 / (4.0 * len);
        const double mx = sx * inv, my = sy * inv, mz = sz * inv;
        for (int i = 0; i < 4 * len; ++i) { out[3 * i] -= mx; out[3 * i + 1] -= my; out[3 * i + 2] -= mz; }
    }
    for (int i = 12 * len; i < 12 * L; ++i) out[i] = 0.0;   // padding residues
}

}  // namespace

extern "C" int e3d_nerf_backbone(const float* angles, const int32_t* lengths, double* coords, int center, int B, int L,
                                 void* stream) {
    E3D_REQUIRE(angles && lengths && coords && B > 0 && L > 0, "nerf_backbone: bad arguments");
    hipLaunchKernelGGL(nerf_backbone_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, angles, lengths,
                       coords, center, B, L);
    return e3d_launch_status("e3d_nerf_backbone");
}
