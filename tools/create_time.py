import sys, gzip, json, time
sys.path[:0]=[__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))),'zk-email-verify_amd','py')]
import zkwg
base=__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))),'artifacts','o0_ev_1024_1536')
t=time.time(); meta=json.load(open(base+'.json')); sym=gzip.open(base+'.sym.gz','rb').read(); r1cs=gzip.open(base+'.r1cs.gz','rb').read(); print('read', time.time()-t, len(sym)>>20, len(r1cs)>>20)
t=time.time()
c=zkwg.Circuit(zkwg.MAIN_EMAIL_VERIFIER,max_header=1024,max_body=1536,device=-1,sym=sym,sym_alias=meta['alias'],r1cs=r1cs)
print('create', time.time()-t)
