// CPU model of the wave-parallel DDA round (64 steps) vs the serial caster advance.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
struct Dda { int cx,cy,cz,sx,sy,sz; float tx,ty,tz,dx,dy,dz; int steps; };
static int signum(float v){return (0.0f<v)-(v<0.0f);}
static void setup(Dda& d, const float s[3], const float e[3]) {
  const float kEps=1e-6f;
  d.cx=(int)floorf(s[0]+kEps); d.cy=(int)floorf(s[1]+kEps); d.cz=(int)floorf(s[2]+kEps);
  int ex=(int)floorf(e[0]+kEps), ey=(int)floorf(e[1]+kEps), ez=(int)floorf(e[2]+kEps);
  d.steps=abs(ex-d.cx)+abs(ey-d.cy)+abs(ez-d.cz);
  float r[3]={e[0]-s[0],e[1]-s[1],e[2]-s[2]};
  d.sx=signum(r[0]); d.sy=signum(r[1]); d.sz=signum(r[2]);
  float sh[3]={s[0]-(float)d.cx,s[1]-(float)d.cy,s[2]-(float)d.cz};
  float b[3]={(float)(d.sx>0)-sh[0],(float)(d.sy>0)-sh[1],(float)(d.sz>0)-sh[2]};
  d.tx=b[0]/r[0]; d.ty=b[1]/r[1]; d.tz=b[2]/r[2];
  d.dx=(float)d.sx/r[0]; d.dy=(float)d.sy/r[1]; d.dz=(float)d.sz/r[2];
}
static void advance(Dda& d){ bool y_lt=d.ty<d.tx; float m=y_lt?d.ty:d.tx; bool z_lt=d.tz<m;
  if(z_lt){d.cz+=d.sz;d.tz+=d.dz;} else if(y_lt){d.cy+=d.sy;d.ty+=d.dy;} else {d.cx+=d.sx;d.tx+=d.dx;} }
static int count_lt(const float* e, float v){ const float* base=e; int n=64; while(n>1){int h=n/2; base=(base[h-1]<v)?base+h:base; n-=h;} return (int)(base-e)+(base[0]<v); }
static int count_le(const float* e, float v){ const float* base=e; int n=64; while(n>1){int h=n/2; base=(base[h-1]<=v)?base+h:base; n-=h;} return (int)(base-e)+(base[0]<=v); }
static bool eligible(const Dda& d){ return d.sx&&d.sy&&d.sz&&std::isfinite(d.tx)&&std::isfinite(d.ty)&&std::isfinite(d.tz)&&std::isfinite(d.dx)&&std::isfinite(d.dy)&&std::isfinite(d.dz); }
// one round: voxels of the next 64 steps into vox[64][3]; state advanced by 64 events
static void round64(Dda& st, int vox[64][3]){
  float E[3][65];
  for(int a=0;a<3;++a){ float t=a==0?st.tx:a==1?st.ty:st.tz, d=a==0?st.dx:a==1?st.dy:st.dz; for(int i=0;i<65;++i){E[a][i]=t; t=t+d;} }
  int n[3]={0,0,0};
  for(int lane=0;lane<64;++lane){
    float ex=E[0][lane],ey=E[1][lane],ez=E[2][lane];
    int cyx=count_lt(E[1],ex), czx=count_lt(E[2],ex);
    int cxy=count_le(E[0],ey), czy=count_lt(E[2],ey);
    int cxz=count_le(E[0],ez), cyz=count_le(E[1],ez);
    int rx=lane+cyx+czx, ry=lane+cxy+czy, rz=lane+cxz+cyz;
    if(rx<64){vox[rx][0]=st.cx+st.sx*lane; vox[rx][1]=st.cy+st.sy*cyx; vox[rx][2]=st.cz+st.sz*czx; n[0]++;}
    if(ry<64){vox[ry][0]=st.cx+st.sx*cxy; vox[ry][1]=st.cy+st.sy*lane; vox[ry][2]=st.cz+st.sz*czy; n[1]++;}
    if(rz<64){vox[rz][0]=st.cx+st.sx*cxz; vox[rz][1]=st.cy+st.sy*cyz; vox[rz][2]=st.cz+st.sz*lane; n[2]++;}
  }
  st.cx+=st.sx*n[0]; st.cy+=st.sy*n[1]; st.cz+=st.sz*n[2];
  st.tx=E[0][n[0]]; st.ty=E[1][n[1]]; st.tz=E[2][n[2]];
}
int main(){
  std::mt19937_64 rng(7); std::uniform_real_distribution<float> U(-100.f,100.f), L(0.f,1.f);
  long rays=0, bad=0, inel=0, ties=0;
  for(long it=0; it<400000; ++it){
    float s[3],e[3];
    int kind=it%8;
    for(int k=0;k<3;++k){ s[k]=U(rng); e[k]=U(rng); }
    if(kind==1){ for(int k=0;k<3;++k){ s[k]=floorf(s[k])+0.5f; e[k]=s[k]+(float)((int)(L(rng)*200)-100); } }   // diagonal-ish lattice directions: many ties
    if(kind==2){ float dlt=(float)((int)(L(rng)*150)+1); for(int k=0;k<3;++k){ s[k]=floorf(s[k])+0.5f; e[k]=s[k]+((rng()&1)?dlt:-dlt);} } // exact diagonal: triple ties
    if(kind==3){ for(int k=0;k<3;++k){ s[k]=floorf(s[k])+0.25f; e[k]=s[k]+(float)((int)(L(rng)*64)-32)*0.5f; } }
    if(kind==4){ e[0]=s[0]; }  // zero component -> ineligible
    if(kind==5){ for(int k=0;k<3;++k){ s[k]=floorf(s[k]); e[k]=floorf(e[k]); } }  // starts on lattice corners
    Dda a; setup(a,s,e); Dda b=a;
    if(!eligible(a)){ ++inel; continue; }
    ++rays;
    int total=a.steps+1;
    for(int s0=0; s0<total; s0+=64){
      int vox[64][3]; Dda pre=b; round64(b,vox);
      int nr = total-s0<64?total-s0:64;
      for(int i=0;i<nr;++i){
        if(vox[i][0]!=a.cx||vox[i][1]!=a.cy||vox[i][2]!=a.cz){ if(bad<5) printf("MISMATCH it %ld step %d: serial (%d %d %d) parallel (%d %d %d)\n",it,s0+i,a.cx,a.cy,a.cz,vox[i][0],vox[i][1],vox[i][2]); ++bad; break; }
        if(a.tx==a.ty||a.ty==a.tz||a.tx==a.tz) ++ties;
        advance(a);
      }
      if(nr==64){ // state check after a full round
        if(a.cx!=b.cx||a.cy!=b.cy||a.cz!=b.cz||memcmp(&a.tx,&b.tx,4)||memcmp(&a.ty,&b.ty,4)||memcmp(&a.tz,&b.tz,4)){ if(bad<5) printf("STATE MISMATCH it %ld s0 %d\n",it,s0); ++bad; break; }
      }
      (void)pre;
    }
  }
  printf("rays %ld ineligible %ld ties seen %ld bad %ld\n",rays,inel,ties,bad);
  return bad!=0;
}
