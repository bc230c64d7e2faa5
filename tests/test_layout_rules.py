"""CPU: who may touch the oracle.  Only tests/ (incl. tests/diag/), __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _py_files(d):
    for base, _, files in os.walk(d):
        if '__pycache__' in base:
            continue
        for f in files:
            if f.endswith(('.py', '.sh')):
                yield os.path.join(base, f)


def test_only_test_infrastructure_imports_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for d in ('spatial-intention-maps_amd', 'tools'):
        for f in _py_files(os.path.join(ROOT, d)):
            assert not pat.search(open(f).read()), '%s imports the oracle' % f
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    # bench.py: the import sits inside cpu_baseline() only
    for m in pat.finditer(bench):
        head = bench[:m.start()]
        assert head.rfind('def cpu_baseline') > head.rfind('\ndef main'), 'bench.py imports the oracle outside cpu_baseline()'
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert 'oracle' in entry      # build() imports it as its "build", smoke() uses it as the checker
