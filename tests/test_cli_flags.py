"""The re-hosted command lines (scripts/train.py, scripts/test.py,
scripts/multifuture_inference.py) against the reference's own argparse tables:
every flag the reference declares exists here with the same type, default and
action, read from the reference SOURCE with `ast` (nothing is imported or executed).
Needs the /root/reference checkout; skipped elsewhere (the GPU box)."""
import ast
import os

import pytest

from multiverse_amd import cli

REF = "/root/reference/code"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")


def _literal(node):
  try:
    return ast.literal_eval(node)
  except Exception:   # names like int / str / float
    return getattr(node, "id", None)


def reference_flags(path):
  """{flag: dict(type=, default=, action=)} of every parser.add_argument call."""
  tree = ast.parse(open(path).read())
  out = {}
  for node in ast.walk(tree):
    if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute)
            and node.func.attr == "add_argument" and node.args):
      continue
    name = _literal(node.args[0])
    if not isinstance(name, str):
      continue
    kw = {k.arg: _literal(k.value) for k in node.keywords}
    out[name] = {"type": kw.get("type"), "default": kw.get("default"),
                 "action": kw.get("action")}
  return out


def ours(parser):
  out = {}
  for a in parser._actions:
    for opt in (a.option_strings or [a.dest]):
      typ = getattr(a.type, "__name__", None) if a.type else None
      action = "store_true" if a.__class__.__name__ == "_StoreTrueAction" else None
      out[opt] = {"type": typ, "default": a.default, "action": action}
  return out


@pytest.mark.parametrize("script,kind", [("train.py", "t"), ("test.py", "T")])
def test_model_scripts_declare_the_reference_flags(script, kind):
  ref = reference_flags(os.path.join(REF, script))
  mine = ours(cli.model_parser(kind))
  assert len(ref) > 40
  missing = sorted(set(ref) - set(mine))
  assert not missing, missing
  for flag, r in ref.items():
    m = mine[flag]
    if r["action"] == "store_true":
      assert m["action"] == "store_true" and m["default"] in (False, None), flag
      continue
    assert m["type"] == (r["type"] or "str") or (r["type"] is None and m["type"] is None), \
        (flag, r, m)
    assert m["default"] == r["default"], (flag, r["default"], m["default"])
  extra = sorted(set(mine) - set(ref) - {"-h", "--help"})
  assert extra == ["--compact_inputs"], extra      # the one flag that is not the reference's


def test_multifuture_inference_declares_the_reference_flags():
  ref = reference_flags(os.path.join(REF, "multifuture_inference.py"))
  mine = ours(cli.multifuture_inference_parser())
  missing = sorted(set(ref) - set(mine))
  assert not missing, missing
  for flag, r in ref.items():
    if r["action"] == "store_true":
      assert mine[flag]["action"] == "store_true", flag
    else:
      assert mine[flag]["default"] == r["default"], (flag, r["default"], mine[flag]["default"])
