#include "../mallie_amd/csrc/mgpu_device.hpp"
#include <cstdio>
using namespace mgpu;
__global__ void k(double nx, double ny, double nz, uint32_t a, uint32_t b, uint32_t c, uint32_t d, double* out) {
  Rng r{a,b,c,d};
  V3 n = v3(nx,ny,nz);
  V3 sd = sample_diffuse(n, r);
  out[0]=sd.x; out[1]=sd.y; out[2]=sd.z;
  Rng r2{a,b,c,d};
  double u1 = rng_next(r2), u2 = rng_next(r2);
  double theta = acos(sqrt(1.0-u1)); double phi = 6.283185307179586*u2;
  out[3]=u1; out[4]=u2; out[5]=theta; out[6]=phi; out[7]=cos(theta); out[8]=sin(theta); out[9]=cos(phi); out[10]=sin(phi);
}
int main(){
  double* d; hipMalloc(&d, 8*16);
  uint32_t st[4]; hash_state(42,0,3*96+69,st);
  // advance 3 draws on host: 2 jitter + unused r
  auto next=[&](){ uint32_t t=st[0]^(st[0]<<11); st[0]=st[1];st[1]=st[2];st[2]=st[3]; st[3]=(st[3]^(st[3]>>19))^(t^(t>>8)); return st[3]*(1.0/4294967296.0); };
  next();next();next();
  k<<<1,1>>>(-0.9999832173877716,-0.005793525938556732,0.0, st[0],st[1],st[2],st[3], d);
  double h[16]; hipMemcpy(h,d,8*16,hipMemcpyDeviceToHost);
  printf("dev sd = %.17g %.17g %.17g\n", h[0],h[1],h[2]);
  printf("dev u1=%.17g u2=%.17g theta=%.17g phi=%.17g cos_t=%.17g sin_t=%.17g cos_p=%.17g sin_p=%.17g\n",h[3],h[4],h[5],h[6],h[7],h[8],h[9],h[10]);
  double u1=next(), u2=next(); double theta=acos(sqrt(1.0-u1)), phi=6.283185307179586*u2;
  printf("hst u1=%.17g u2=%.17g theta=%.17g phi=%.17g cos_t=%.17g sin_t=%.17g cos_p=%.17g sin_p=%.17g\n",u1,u2,theta,phi,cos(theta),sin(theta),cos(phi),sin(phi));
  return 0;
}
