// measures the self-synchronisation distance (in 1024-bit sub-sequences) of speculative JPEG Huffman decoding
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "jpeg_synth.h"
typedef struct { uint16_t lut[65536]; } Tab;  // 16-bit direct: len<<8|sym
static void build(Tab* t, const uint8_t* counts, const uint8_t* vals){ memset(t,0,sizeof *t); unsigned code=0,k=0; for(int l=1;l<=16;l++){ for(int i=0;i<counts[l-1];i++,k++){ unsigned lo=code<<(16-l), n=1u<<(16-l); for(unsigned j=0;j<n;j++) t->lut[lo+j]=(l<<8)|vals[k]; code++; } code<<=1; } }
static uint8_t* U; static size_t UL; // unstuffed
static inline unsigned peek16(size_t p){ size_t b=p>>3; unsigned v=(U[b]<<24)|(U[b+1]<<16)|(U[b+2]<<8)|U[b+3]; return (v<<(p&7))>>16; }
typedef struct { size_t p; int c,k; } St;
static Tab T[4]; static int nb; static int slot_of[10];
static void step(St* s){ // one symbol
  Tab* t=&T[slot_of[s->c]*2+(s->k?1:0)]; unsigned e=t->lut[peek16(s->p)]; unsigned len=e>>8, sym=e&255; if(!len){ s->p+=1; return; }
  s->p+=len+(sym&15); int done=0; if(s->k==0){ s->k=1; } else if(sym==0) done=1; else { s->k+= (sym>>4)+1; if(s->k>=64) done=1; }
  if(done){ s->k=0; s->c=(s->c+1)%nb; } }
int main(int argc,char**argv){ int hs=argc>1?atoi(argv[1]):2, vs=argc>2?atoi(argv[2]):2, S=argc>3?atoi(argv[3]):1024, opt=argc>4?atoi(argv[4]):0;
  JsynthParams p={1920,1080,hs,vs,85,0,0,opt,0,12,11}; size_t cap=8<<20; uint8_t* f=malloc(cap); size_t n=jsynth_encode(&p,f,cap);
  // parse DHT + find SOS
  size_t pos=2, ss=0; uint8_t cnt[4][16], val[4][256]; while(pos<n){ unsigned m=f[pos+1]; unsigned len=(f[pos+2]<<8)|f[pos+3]; if(m==0xC4){ size_t q=pos+4; while(q<pos+2+len){ int tc=f[q]>>4, th=f[q]&15; int id=th*2+tc; memcpy(cnt[id],f+q+1,16); int tot=0; for(int i=0;i<16;i++) tot+=cnt[id][i]; memcpy(val[id],f+q+17,tot); q+=17+tot; } } if(m==0xDA){ ss=pos+2+len; break; } pos+=2+len; }
  for(int i=0;i<4;i++) build(&T[i],cnt[i],val[i]);
  nb=hs*vs+2; for(int i=0;i<hs*vs;i++) slot_of[i]=0; slot_of[hs*vs]=1; slot_of[hs*vs+1]=1;
  U=malloc(n+16); UL=0; for(size_t i=ss;i+1<n;i++){ if(f[i]==0xFF && f[i+1]==0xD9) break; U[UL++]=f[i]; if(f[i]==0xFF && f[i+1]==0) i++; } memset(U+UL,0,16);
  size_t nsub=(UL*8+S-1)/S; St* truth=malloc(sizeof(St)*(nsub+1)); St s={0,0,0}; for(size_t i=0;i<nsub;i++){ while(s.p<(i+1)*(size_t)S && s.p<UL*8) step(&s); truth[i]=s; }
  if(argc>5){ FILE* fo=fopen(argv[5],"wb"); for(size_t i=0;i<nsub;i++){ uint32_t r[3]={(uint32_t)truth[i].p,(uint32_t)truth[i].c,(uint32_t)truth[i].k}; fwrite(r,4,3,fo);} fclose(fo); }
  // speculative chains
  long hist[12]={0}; long maxd=0; double sum=0; for(size_t i=1;i<nsub;i+=1){ St q={i*(size_t)S,0,0}; size_t j=i; for(;j<nsub;j++){ while(q.p<(j+1)*(size_t)S && q.p<UL*8) step(&q); if(q.p==truth[j].p&&q.c==truth[j].c&&q.k==truth[j].k) break; } long d=j-i+1; sum+=d; if(d>maxd)maxd=d; int b=0; long x=d; while(x>1){x>>=1;b++;} hist[b>11?11:b]++; }
  printf("hs%d vs%d S=%d opt=%d nsub=%zu meanD=%.1f maxD=%ld hist(log2):",hs,vs,S,opt,nsub,sum/(nsub-1),maxd); for(int i=0;i<12;i++) printf(" %ld",hist[i]); printf("\n"); return 0; }
