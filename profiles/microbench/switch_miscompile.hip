#include "../../mallie_amd/csrc/mgpu_device.hpp"
#include <cstdio>
using namespace mgpu;
__global__ void k(const double* nin, double* out) {
  V3 n = v3(nin[0],nin[1],nin[2]);
  const double ax = (double)fabsf((float)n.x), ay = (double)fabsf((float)n.y), az = (double)fabsf((float)n.z);
  int index = 0;
  double minval = ax;
  if (!(ax < 1.0e+6)) { index = -1; minval = 1.0e+6; }
  if (ay < minval) { minval = ay; index = 1; }
  if (az < minval) { minval = az; index = 2; }
  V3 t;
  if (index == 0) t = v3(0.0, -n.z, n.y);
  else if (index == 1) t = v3(-n.z, 0.0, n.x);
  else t = v3(-n.y, n.x, 0.0);
  out[0]=index; out[1]=t.x; out[2]=t.y; out[3]=t.z; out[4]=ax; out[5]=ay; out[6]=az;
  t = normalized(t);
  out[7]=t.x; out[8]=t.y; out[9]=t.z;
  V3 b = normalized(cross(t, n));
  out[10]=b.x; out[11]=b.y; out[12]=b.z;
}
int main(){
  double* d; double *din; hipMalloc(&d, 8*16); hipMalloc(&din, 24);
  double nh[3] = {-0.9999832173877716,-0.005793525938556732,0.0};
  hipMemcpy(din, nh, 24, hipMemcpyHostToDevice);
  k<<<1,1>>>(din, d);
  double h[16]; hipMemcpy(h,d,8*16,hipMemcpyDeviceToHost);
  printf("index=%g t=(%.17g %.17g %.17g) a=(%g %g %g)\n tn=(%.17g %.17g %.17g)\n b=(%.17g %.17g %.17g)\n",h[0],h[1],h[2],h[3],h[4],h[5],h[6],h[7],h[8],h[9],h[10],h[11],h[12]);
  return 0;
}
