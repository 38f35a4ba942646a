"""CMU mocap `.amc` files: reader and writer.

Mirrors /root/reference/lib/mocap_data.py:41-109 (`subject_trial_path`,
`parse_amc_data`, `amc_to_array`, `load_mocap_trial`) and the writer the mocap
script carries inline (/root/reference/mo_mo_mocap_oivae.py:328-351).  The
reference parses with a `lark` LALR grammar (lib/mocap_data.py:10-39); this is a
hand-written scanner for the same language:

    start:       preamble keyframe*
    preamble:    ":FULLY-SPECIFIED" ":DEGREES"
    keyframe:    INT measurement+
    measurement: WORD " " (SIGNED_NUMBER " ")* SIGNED_NUMBER
    '#' comments and newlines are ignored

Parity note: `lark` is not installed in the build container, so the reference's
parser could not be run; the behaviour is pinned by the grammar above and by
round trips through the writer (tests/test_data_formats.py) -- parity unpinned
against the reference's own parse results.
"""
import os
import re

DATADIR = os.path.join(os.getcwd(), 'data')

_WORD = re.compile(r"[A-Za-z]+\Z")
_INT = re.compile(r"[0-9]+\Z")
_NUM = re.compile(r"[+-]?(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?\Z")


def subject_trial_path(subject, trial, datadir=DATADIR):
  """<datadir>/cmu_mocap/subjects/SS/SS_TT.amc, the layout the reference reads (lib/mocap_data.py:41-53)."""
  ss, tt = "%02d" % subject, "%02d" % trial
  return os.path.join(datadir, "cmu_mocap", "subjects", ss, "%s_%s.amc" % (ss, tt))


def parse_amc_data(raw_contents):
  """String contents of an .amc file -> list of (key frame id, {joint: [channels]})
  (lib/mocap_data.py:55-60).  Raises ValueError on text outside the grammar."""
  lines = []
  for ln, raw in enumerate(raw_contents.split('\n'), 1):
    text = raw.split('#', 1)[0].strip()
    if text:
      lines.append((ln, text))
  if len(lines) < 2 or lines[0][1] != ':FULLY-SPECIFIED' or lines[1][1] != ':DEGREES':
    # the two preamble tokens may also share a line
    if not lines or lines[0][1].split() != [':FULLY-SPECIFIED', ':DEGREES']:
      raise ValueError("amc: expected the preamble ':FULLY-SPECIFIED' ':DEGREES'")
    body = lines[1:]
  else:
    body = lines[2:]
  frames = []
  for ln, text in body:
    if _INT.match(text):
      frames.append((int(text), {}))
      continue
    toks = text.split()
    if not frames:
      raise ValueError("amc line %d: measurement before the first key frame" % ln)
    if len(toks) < 2 or not _WORD.match(toks[0]) or not all(_NUM.match(t) for t in toks[1:]):
      raise ValueError("amc line %d: expected 'joint number [number ...]', got %r" % (ln, text))
    frames[-1][1][toks[0]] = [float(t) for t in toks[1:]]
  for fid, kvs in frames:
    if not kvs:
      raise ValueError("amc: key frame %d has no measurements" % fid)
  return frames


def amc_to_array(parsed_amc, joint_order):
  """Parsed frames -> (joint_dims {joint: channel count}, rows: one flat channel list per frame, joints in
  `joint_order`).  Rejects (AssertionError, as the reference does, lib/mocap_data.py:62-87) frame ids that are
  not 1, 2, 3, ..., frames whose joint set differs from `joint_order`, and joints whose channel count changes."""
  wanted = sorted(joint_order)
  first = parsed_amc[0][1]
  joint_dims = {joint: len(first[joint]) for joint in joint_order}
  rows = []
  for expect_id, (frame_id, channels) in enumerate(parsed_amc, 1):
    assert frame_id == expect_id, "amc: frame ids must count from 1 (got %d at position %d)" % (frame_id, expect_id)
    assert sorted(channels) == wanted, "amc: frame %d holds other joints than joint_order" % frame_id
    row = []
    for joint in joint_order:
      values = channels[joint]
      assert len(values) == joint_dims[joint], "amc: frame %d: %s has %d channels" % (frame_id, joint, len(values))
      row.extend(values)
    rows.append(row)
  return joint_dims, rows


def load_mocap_trial(subject, trial, joint_order=None, datadir=DATADIR):
  """One trial of one subject -> (joint_order, joint_dims, rows); without a given order: root first, then
  the other joints as the first frame lists them (lib/mocap_data.py:89-109)."""
  with open(subject_trial_path(subject, trial, datadir=datadir), "r") as f:
    frames = parse_amc_data(f.read())
  if joint_order is None:
    joint_order = ["root"] + [joint for joint in frames[0][1] if joint != "root"]
  joint_dims, rows = amc_to_array(frames, joint_order)
  return joint_order, joint_dims, rows


def save_amc(filename, seq, joint_order, joint_dims):
  """Write poses [T, sum dims] as an .amc file (mo_mo_mocap_oivae.py:328-351).  As in
  the script, `joint_dims['root']` counts only the learned rotation channels and the
  three root translation channels are written as zeros."""
  ordered_joint_dims = [joint_dims[joint] for joint in joint_order]
  starts = [0]
  for d in ordered_joint_dims:
    starts.append(starts[-1] + d)
  with open(filename, 'w') as f:
    f.write('#!OML:ASF\n')
    f.write(':FULLY-SPECIFIED\n')
    f.write(':DEGREES\n')
    for i in range(len(seq)):
      f.write('{}\n'.format(i + 1))
      for joint, dim, ix_start in zip(joint_order, ordered_joint_dims, starts):
        f.write('{} '.format(joint))
        if joint == 'root':
          f.write('0 0 0 ')
        f.write(' '.join(['%.9g' % float(seq[i][ix_start + j]) for j in range(dim)]))
        f.write('\n')
