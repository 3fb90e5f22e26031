// tools/parse_stats.cpp — statistics of the fastEncoder parse on the synthetic corpora (analysis only; not part of the product):
// probes, sequences, match-length distribution, and how often the same-hash predecessor of a position was inserted.
//   g++ -O2 -o tools/_build/parse_stats tools/parse_stats.cpp compress_amd/csrc/kc_corpus.cpp -lpthread
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../include/kcgpu.h"
static inline uint64_t ld64(const uint8_t*p){uint64_t v;memcpy(&v,p,8);return v;}
static inline uint32_t ld32(const uint8_t*p){uint32_t v;memcpy(&v,p,4);return v;}
static inline uint32_t h6(uint64_t u){return (uint32_t)(((u<<16)*227718039650203ULL)>>(64-15));}
int main(int argc,char**argv){
  int kind = argc>1? argv[1][0]:'T';
  const int U=131072; int NU=32;
  std::vector<uint8_t> buf((size_t)U*NU);
  kc_corpus_fill(kind, kind=='T'?0x5EED0001: kind=='H'?0x5EED0002:kind=='J'?0x5EED0003:0x5EED0004,0,NU,U,buf.data(),4);
  long probes=0,seqs=0,rep=0,c0=0,c1=0,o2m=0, lookups=0, predIns=0, chainSteps=0, noPred=0, candValid=0, candEq=0, mlsum=0, llsum=0;
  long depthHist[8]={0}; long fl[40]={0}, bk[12]={0};
  long bs_runs=0; long backsum=0;
  for(int u=0;u<NU;u++){
    const uint8_t*src=buf.data()+(size_t)u*U;
    std::vector<int> table(32768,-1);
    std::vector<int> pred(U,-1); std::vector<int> last(32768,-1);
    for(int p=0;p+8<=U;p++){uint32_t h=h6(ld64(src+p));pred[p]=last[h];last[h]=p;}
    std::vector<char> ins(U,0);
    auto lookup=[&](int p,uint32_t h){ lookups++; int q=pred[p]; int d=0; while(q>=0&&!ins[q]){q=pred[q];d++;} 
       if(q!=table[h]){fprintf(stderr,"chain mismatch p=%d q=%d tab=%d\n",p,q,table[h]);}
       if(q<0)noPred++; depthHist[std::min(d,7)]++; chainSteps+=d; return q;};
    // blocks of 64K with history
    int o1=1,o2=4; int nseq_total=0;
    for(int b=0;b<2;b++){
      int blkStart=b*65536, blkEnd=blkStart+65536; int sLimit=blkEnd-8; int s=blkStart,nextEmit=s; int nseq=0;
      // Encode vs EncodeNoHist differences ignored (canRepeat snapshot)
      for(;;){
        int t=0; bool canRepeat=nseq>2; bool done=false;
        for(;;){
          uint64_t cv=ld64(src+s); uint32_t nh=h6(cv), nh2=h6(cv>>8);
          probes++;
          int cand=lookup(s,nh); 
          // candidate2: table state before writes
          int cand2;{ lookups++; int q=pred[s+1]; int d=0; while(q>=0&&(!ins[q])){q=pred[q];d++;} if(q!=table[nh2])fprintf(stderr,"chain2 mismatch\n"); if(q<0)noPred++; depthHist[std::min(d,7)]++; chainSteps+=d; cand2=q;}
          int repIndex=s-o1+2;
          table[nh]=s; table[nh2]=s+1; ins[s]=1; ins[s+1]=1;
          if(canRepeat&&repIndex>=0&&ld32(src+repIndex)==(uint32_t)(cv>>16)){
            int len=4; while(s+2+len<blkEnd && src[s+2+len]==src[repIndex+len])len++;
            int start=s+2; while(repIndex>0&&start>nextEmit+1&&src[repIndex-1]==src[start-1]){repIndex--;start--;len++;backsum++;}
            rep++; seqs++; nseq++; mlsum+=len; llsum+=start-nextEmit;
            s+=len - (s+2-start) +2; // s += length+2 where length excludes back
            nextEmit=s; if(s>=sLimit){done=true;break;} continue;
          }
          if(cand>=0){candValid++; if(ld32(src+cand)==(uint32_t)cv){candEq++;}}
          if(cand>=0&&ld32(src+cand)==(uint32_t)cv){t=cand;c0++;break;}
          if(cand2>=0&&ld32(src+cand2)==(uint32_t)(cv>>8)){t=cand2;s++;c1++;break;}
          s+=2+((s-nextEmit)>>5); if(s>=sLimit){done=true;break;}
        }
        if(done)break;
        o2=o1;o1=s-t; int l=4; while(s+l<blkEnd&&src[s+l]==src[t+l])l++; fl[l<39?l:39]++; {int bb=0; int ss=s,tt=t; while(tt>0&&ss>nextEmit&&src[tt-1]==src[ss-1]){ss--;tt--;bb++;} bk[bb<11?bb:11]++;}
        while(t>0&&s>nextEmit&&src[t-1]==src[s-1]){s--;t--;l++;backsum++;}
        seqs++;nseq++; mlsum+=l; llsum+=s-nextEmit; s+=l; nextEmit=s; if(s>=sLimit)break;
        uint64_t cv=ld64(src+s); int p2=s-o2;
        if(canRepeat&&ld32(src+p2)==(uint32_t)cv){int l2=4; while(s+l2<blkEnd&&src[s+l2]==src[p2+l2])l2++; uint32_t nh=h6(cv); table[nh]=s; ins[s]=1; o2m++; seqs++;nseq++; mlsum+=l2; s+=l2; nextEmit=s; std::swap(o1,o2); if(s>=sLimit)break;}
      }
    }
  }
  printf("kind %c per unit: probes %.0f seqs %.0f (rep %.0f c0 %.0f c1 %.0f o2 %.0f) lookups %.0f\n",kind,(double)probes/NU,(double)seqs/NU,(double)rep/NU,(double)c0/NU,(double)c1/NU,(double)o2m/NU,(double)lookups/NU);
  printf("avg ml %.2f avg ll %.2f back/seq %.2f\n",(double)mlsum/seqs,(double)llsum/seqs,(double)backsum/seqs);
  printf("chain: noPred %.3f avg extra steps %.3f; depth hist:",(double)noPred/lookups,(double)chainSteps/lookups);
  for(int i=0;i<8;i++)printf(" %.3f",(double)depthHist[i]/lookups); printf("\n");
  printf("fwd len cdf:"); {long tot=0,c=0; for(int i=0;i<40;i++)tot+=fl[i]; for(int i=0;i<40;i++){c+=fl[i]; if(i==7||i==11||i==15||i==19||i==23||i==31||i==39)printf(" <=%d:%.3f",i,(double)c/tot);} printf("\nback hist:"); long tb=0; for(int i=0;i<12;i++)tb+=bk[i]; for(int i=0;i<12;i++)printf(" %.3f",(double)bk[i]/tb); printf("\n");}
  printf("cand0 valid %.3f of probes, eq4 %.3f of valid\n",(double)candValid/probes,(double)candEq/candValid);
}
