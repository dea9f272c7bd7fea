python -m pytest tests/test_longseq.py -x -q -m gpu --durations=8 -k "python_host_byte_identical_over_the_pieces and long_six or cpp_host_byte_identical_over_the_pieces and long_six" 2>&1 | tail -20
mkdir -p /tmp/l && cd /tmp/l && python - <<'PY'
import lzma, gzip, shutil, os, subprocess, time
S='/root/repo/tests/golden/batches/'
open('long.fasta','wb').write(lzma.open(S+'long.fasta.xz').read())
for f in ("content.txt.gz","idx_f.txt.gz"): open(f[:-3],'wb').write(gzip.open(S+f).read())
for f in ("idx","idx_info.txt","idx_trie","idx_trie.txt"): shutil.copy(S+f,f)
t=time.time()
r=subprocess.run(['/root/repo/kasa_amd/host/kasa_identify','identify','-c','content.txt','-d','idx','-i','long.fasta','-q','o.jsonl','-p','p.csv','--jsonl','-b','100','-m','1','-n','1','-v','--six'],env=dict(os.environ,KASA_HOST_TIMING='1'),stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
print(time.time()-t); print(r.stdout[-3000:])
PY
