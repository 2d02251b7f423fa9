"""integration/orcvio_msckf.patch: the call-site patch INTEGRATION.md describes, as a file that applies.  In the build
container the reference tree is at /root/reference: `git apply --check` must accept the patch there (the anchors -- src/orcvio.cpp
:2154-2193 removeLostObjects, :2497-2560 removeLostFeatures, :2803-2851 pruneImuStateBuffer, include/orcvio/orcvio.h:200-214,
CMakeLists.txt -- still match the reference text).  The patch travels, the reference tree does not: skipped where it is absent.
Wherever it is: everything the patch adds to the C++ sources sits inside #ifdef ORCVIO_USE_AMD_MSCKF, so the reference builds and
behaves as before without the definition, and it names only entry points include/orcvio_msckf.h declares."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, 'integration', 'orcvio_msckf.patch')
REF = '/root/reference'


def _hunks():
    files, cur = {}, None
    for line in open(PATCH).read().splitlines():
        if line.startswith('+++ '):
            cur = line[4:].split('\t')[0]
            cur = cur[2:] if cur.startswith('b/') else cur
            files[cur] = []
        elif line.startswith('--- ') or line.startswith('@@'):
            if cur is not None and line.startswith('@@'):
                files[cur].append([])
        elif cur is not None and files[cur]:
            files[cur][-1].append(line)
    return files


def test_patch_touches_the_three_call_sites_and_nothing_else():
    files = _hunks()
    assert set(files) == {'CMakeLists.txt', 'include/orcvio/orcvio.h', 'src/orcvio.cpp'}
    added = '\n'.join(l[1:] for h in files['src/orcvio.cpp'] for l in h if l.startswith('+'))
    for name in ('gpuFeatureUpdate(msckf_feature_ids, nullptr)', 'gpuFeatureUpdate(used_IDs, &rm_imu_state_ids)', 'gpuObjectUpdate(H_x, H_f, res)'):
        assert name in added, name
    # no line of the reference is removed: the reference path stays, behind the #else
    assert not any(l.startswith('-') for hs in files.values() for h in hs for l in h if isinstance(l, str) and len(l) > 0)


def test_everything_added_to_the_sources_is_behind_the_definition():
    files = _hunks()
    for f in ('include/orcvio/orcvio.h', 'src/orcvio.cpp'):
        for h in files[f]:
            depth = 0
            for l in h:
                if not l.startswith('+'):
                    assert depth == 0 or l.strip() in ('', ' '), (f, l)   # (context lines sit outside the guarded blocks)
                    continue
                t = l[1:].strip()
                if t.startswith('#ifdef ORCVIO_USE_AMD_MSCKF'):
                    depth += 1
                elif t.startswith('#endif'):
                    depth -= 1
                    assert depth >= 0
                else:
                    assert depth > 0 or t == '', (f, l)
            assert depth == 0, f


def test_patch_calls_only_declared_entry_points():
    header = open(os.path.join(ROOT, 'include', 'orcvio_msckf.h')).read()
    declared = set(re.findall(r'\b(orcvio_msckf_\w+)\s*\(', header)) | set(re.findall(r'\b(orcvio_msckf_\w+)\b', header))
    used = set(re.findall(r'\b(orcvio_msckf_\w+)\b', '\n'.join(l[1:] for l in open(PATCH).read().splitlines() if l.startswith('+') and not l.startswith('+++'))))
    used -= {'orcvio_msckf_h'}
    lib_names = {u for u in used if not u.endswith('.so')}
    assert lib_names <= declared | {'orcvio_msckf'}, lib_names - declared


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'src')), reason='the reference tree is not on this machine (it never travels to the GPU box)')
def test_patch_applies_to_the_reference_tree():
    r = subprocess.run(['git', 'apply', '--check', '-p1', '--verbose', PATCH], cwd=REF, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
