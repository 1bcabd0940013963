"""The boundary is a C ABI: the four headers under include/ must be valid C (not only C++), and a plain C program must
link against the product library and call it.  CPU only: the host-side entry points (error strings, the Kaiser window
and chirp tables of lib/demod_impl.cc:85,92-101, the channeliser's tap design) run without a GPU; lora_hip_demod_create
must fail with LORA_HIP_ENODEVICE here or succeed on a GPU box -- never anything else."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gr-lora_amd", "liblora_hip_demod.so")

PROGRAM = r"""
#include "lora_hip_demod.h"
#include "lora_demod_core.h"
#include "lora_hip_chan.h"
#include "lora_decode.h"
#include <stdio.h>
#include <string.h>

int main(void) {
  float w[8];
  lora_c32 up[2 * 64], down[2 * 64];
  lora_win_desc d;
  lora_win_run r;
  lora_hip_demod_t *h = NULL;
  int rc;
  memset(&d, 0, sizeof d); memset(&r, 0, sizeof r);
  if (sizeof d != 16 || sizeof r != 24) return 10;                    /* the layouts the kernels read */
  if (lora_hip_make_kaiser(8, 25.0, w) != LORA_HIP_OK) return 11;
  if (lora_hip_make_chirp_tables(6, up, down) != LORA_HIP_OK) return 12;
  if (strlen(lora_hip_strerror(LORA_HIP_EINVAL)) == 0) return 13;
  if (lora_hip_demod_batch(NULL, NULL, 0, NULL, 0, NULL, NULL, 0) != LORA_HIP_EINVAL) return 14;
  if (lora_hip_demod_mark(NULL, 0, 0) != LORA_HIP_EINVAL || lora_hip_demod_wait(NULL, LORA_HIP_MARKS) != LORA_HIP_EINVAL) return 15;
  rc = lora_hip_demod_create(&h, 0, 8, 2, 25.0f, NULL, NULL, NULL);
  if (rc != LORA_HIP_OK && rc != LORA_HIP_ENODEVICE) return 16;
  if (rc == LORA_HIP_OK) lora_hip_demod_destroy(h);
  printf("kaiser[3]=%.6f up[1]=(%.6f,%.6f) create=%d flags=%d,%d\n", (double)w[3], (double)up[1].re, (double)up[1].im, rc,
         (int)LORA_HIP_LANE1, (int)LORA_HIP_ASYNC);
  return 0;
}
"""


@pytest.fixture(scope="module")
def cc():
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    return gcc


def test_headers_are_valid_c99(cc, tmp_path):
    src = tmp_path / "abi.c"
    src.write_text(PROGRAM)
    r = subprocess.run([cc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_a_c_program_links_and_calls_the_library(cc, tmp_path):
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    src = tmp_path / "abi.c"
    src.write_text(PROGRAM)
    exe = tmp_path / "abi"
    r = subprocess.run([cc, "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), LIB,
                        "-Wl,-rpath," + os.path.dirname(LIB), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.startswith("kaiser[3]=0.77786") and "flags=128,8" in r.stdout, r.stdout
