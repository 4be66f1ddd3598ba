// ufield_selftest.hip -- host-side self-test of the unsaturated 29-bit-limb arithmetic (csrc/ufield.cuh): the
// drop-in product, the internal-form product / squaring, the dual product, the unreduced add / sub and the conversions against the
// saturated CIOS reference, for all four fields, on random and edge operands.  Runs on the CPU (the functions are
// __host__ __device__); the GPU code path is covered bit-exactly by tests -m gpu.
#include "../zk-cryptography-research-implementations_amd/csrc/ufield.cuh"
#include "../zk-cryptography-research-implementations_amd/csrc/mle_kernels.cuh"
#include <stdio.h>
using namespace zk;
template <class F> int test(const char* name){
  int bad=0;
  for (int it=0; it<20000; it++){
    Fe<F> a = random_element<F>(1, it), b = random_element<F>(2, it);
    if (it==0){ a=fe_zero<F>(); } if (it==1){ for(int i=0;i<F::N;i++){a.l[i]=F::p(i); b.l[i]=F::p(i);} a.l[0]-=1; b.l[0]-=1; }
    if (it==2){ b=fe_one<F>(); }
    Fe<F> want = fe_mul_cios<F>(a,b), got = fe_mul_u<F>(a,b);
    if (!fe_eq<F>(want,got)) { if(bad<3) printf("%s mul mismatch it=%d\n",name,it); bad++; }
    // internal form round trip and products
    Ufe<F> au = u_from_std<F>(a), bu = u_from_std<F>(b);
    Fe<F> back = u_to_std<F>(au);
    if (!fe_eq<F>(back,a)) { if(bad<3) printf("%s roundtrip mismatch it=%d\n",name,it); bad++; }
    Fe<F> p2 = u_to_std<F>(umul<F>(au,bu));
    if (!fe_eq<F>(p2,want)) { if(bad<3) printf("%s umul mismatch it=%d\n",name,it); bad++; }
    // additive: (a+b), (a-b) through unreduced ops then a product by one (internal) to reduce
    Ufe<F> one_u = u_from_std<F>(fe_one<F>());
    Fe<F> s = u_to_std<F>(umul<F>(uadd<F>(au,bu), one_u)), d = u_to_std<F>(umul<F>(usub<F>(au,bu), one_u));
    if (!fe_eq<F>(s, fe_add<F>(a,b)) || !fe_eq<F>(d, fe_sub<F>(a,b))) { if(bad<3) printf("%s addsub mismatch it=%d\n",name,it); bad++; }
    // chained: ((a-b)-2ab... ) style: usub of doubled product
    Ufe<F> q = umul<F>(au,bu); Ufe<F> x = usub<F>(uadd<F>(usqr<F>(au), q), uadd<F>(q,q));
    Fe<F> xs = u_to_std<F>(umul<F>(x, one_u));
    Fe<F> ws = fe_sub<F>(fe_add<F>(fe_sqr<F>(a), want), fe_dbl<F>(want));
    if (!fe_eq<F>(xs, ws)) { if(bad<3) printf("%s chain mismatch it=%d\n",name,it); bad++; }
    // two stored-form products in one scan with one reduction (sparse GKR gate weights): a b + c d, operands up to p - 1 each
    Fe<F> c = random_element<F>(3, it), dd = random_element<F>(4, it);
    if (it==1){ c=a; dd=b; }                                   // (p-1)(p-1) + (p-1)(p-1): the largest sum
    Fe<F> w2 = fe_add<F>(want, fe_mul_cios<F>(c,dd)), g2 = fe_mul2_u<F>(a,b,c,dd);
    if (!fe_eq<F>(w2,g2)) { if(bad<3) printf("%s mul2 mismatch it=%d\n",name,it); bad++; }
    // the uniform-multiplier fold (r4): c + b (dd - c) with b as the pass's challenge; it == 1 is (p-1) + (p-1) * 0 with every operand p - 1,
    // the case whose unreduced value reaches 2 p (the second conditional subtraction of fe_from_u_below_2p)
    if constexpr (UParams<F>::L == 9) {
      UniMul<F> um; unimul_from<F>(um, b);
      Fe<F> wf = fe_add<F>(c, fe_mul_cios<F>(b, fe_sub<F>(dd, c)));
      Fe<F> gf = fe_from_u_below_2p<F>(ufold<F>(um, u_from_limbs32<F>(c), u_from_limbs32<F>(dd)));
      if (!fe_eq<F>(wf,gf)) { if(bad<3) printf("%s ufold mismatch it=%d\n",name,it); bad++; }
      Fe<F> z = fe_zero<F>(), pm = a; if (it==1) {                       // a = p - 1 here
        Fe<F> w3 = fe_add<F>(pm, fe_mul_cios<F>(b, fe_sub<F>(z, pm)));
        Fe<F> g3 = fe_from_u_below_2p<F>(ufold<F>(um, u_from_limbs32<F>(pm), u_from_limbs32<F>(z)));
        if (!fe_eq<F>(w3,g3)) { printf("%s ufold edge mismatch\n",name); bad++; }
      }
    }
  }
  printf("%s: %s\n", name, bad? "FAIL":"ok"); return bad;
}
int main(){ int b=0; b+=test<Fr381>("Fr381"); b+=test<Fq381>("Fq381"); b+=test<Bn254Fq>("Bn254Fq"); b+=test<Bn254Fr>("Bn254Fr"); return b?1:0; }
