#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdint>
int main(){
  for(int p=0;p<4;p++){ float azm = -M_PI/2.0 + (M_PI/4)*p;
    for(int q=0;q<16;q++){ float elv=(M_PI/2.0/16)*q;
      float nx=std::cos(elv)*std::cos(azm), ny=std::cos(elv)*std::sin(azm), nz=std::sin(elv);
      uint32_t a,b,c; memcpy(&a,&nx,4);memcpy(&b,&ny,4);memcpy(&c,&nz,4);
      printf("  {0x%08xu, 0x%08xu, 0x%08xu}, /* p%d q%d */\n",a,b,c,p,q);
    }}
}
