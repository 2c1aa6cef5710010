#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
__global__ void k(const double* x, const double* y, double* o, float* of, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a = x[i], b = y[i];
  o[0*n+i] = sqrt(a);
  o[1*n+i] = 1.0 / b;
  o[2*n+i] = a / b;
  o[3*n+i] = acos(sqrt(1.0 - a));
  o[4*n+i] = cos(6.283185307179586 * a);
  o[5*n+i] = sin(6.283185307179586 * a);
  o[6*n+i] = a * b + a;   // contraction probe
  float fa = (float)a, fb = (float)b;
  of[0*n+i] = -fa / fb;
  of[1*n+i] = sqrtf(fa);
}
static int64_t ulpd(double a, double b){ int64_t ia, ib; memcpy(&ia,&a,8); memcpy(&ib,&b,8); return ia>ib?ia-ib:ib-ia; }
static int64_t ulpf(float a, float b){ int32_t ia, ib; memcpy(&ia,&a,4); memcpy(&ib,&b,4); return ia>ib?ia-ib:ib-ia; }
int main(){
  int n = 1<<20; std::mt19937_64 g(1); std::uniform_real_distribution<double> U(0,1);
  std::vector<double> x(n), y(n); for (int i=0;i<n;i++){ x[i]=U(g); y[i]=(U(g)-0.5)*200; if (y[i]==0) y[i]=1; }
  double *dx,*dy,*dout; float* df; hipMalloc(&dx,8*n); hipMalloc(&dy,8*n); hipMalloc(&dout,8*7*n); hipMalloc(&df,4*2*n);
  hipMemcpy(dx,x.data(),8*n,hipMemcpyHostToDevice); hipMemcpy(dy,y.data(),8*n,hipMemcpyHostToDevice);
  k<<<n/256,256>>>(dx,dy,dout,df,n);
  std::vector<double> o(7*n); std::vector<float> of(2*n);
  hipMemcpy(o.data(),dout,8*7*n,hipMemcpyDeviceToHost); hipMemcpy(of.data(),df,4*2*n,hipMemcpyDeviceToHost);
  const char* names[7]={"sqrt","rcp","div","acos(sqrt(1-a))","cos(2pi a)","sin(2pi a)","a*b+a (no fma)"};
  for (int f=0; f<7; f++){ int64_t mx=0; long ne=0; for(int i=0;i<n;i++){ double a=x[i],b=y[i],r;
      switch(f){case 0:r=sqrt(a);break;case 1:r=1.0/b;break;case 2:r=a/b;break;case 3:r=acos(sqrt(1.0-a));break;case 4:r=cos(6.283185307179586*a);break;case 5:r=sin(6.283185307179586*a);break;default:{ volatile double p=a*b; r=p+a;}}
      int64_t u=ulpd(o[f*n+i],r); if(u>mx)mx=u; if(u)ne++; }
    printf("%-18s max ulp diff %lld, mismatches %ld / %d\n", names[f], (long long)mx, ne, n); }
  { int64_t mx=0; long ne=0; for(int i=0;i<n;i++){ float fa=(float)x[i], fb=(float)y[i]; float r=-fa/fb; int64_t u=ulpf(of[i],r); if(u>mx)mx=u; if(u)ne++; } printf("float div          max ulp diff %lld, mismatches %ld\n",(long long)mx,ne); }
  { int64_t mx=0; long ne=0; for(int i=0;i<n;i++){ float fa=(float)x[i]; float r=sqrtf(fa); int64_t u=ulpf(of[n+i],r); if(u>mx)mx=u; if(u)ne++; } printf("float sqrt         max ulp diff %lld, mismatches %ld\n",(long long)mx,ne); }
  return 0;
}
