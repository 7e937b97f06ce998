/*
 * Driver prelude for the in-place reference build (oracle/_ref).  NOT a stand-in for any missing header:
 * it includes only standard headers, restates the four one-line non-MSVC macros of the reference's
 * template/precomp.h:83-92 (ALIGN, MALLOC64, FREE64, CHECK_RESULT), and then includes the reference's OWN headers
 * from where they lie (-I/root/reference/{template,infra,lib,lib/imgui}).  The reference's full precomp.h
 * cannot be used: it needs <io.h>, "windows.h" and <intrin.h> (template/precomp.h:15,82,223), which this image
 * lacks; writing substitutes for those is not allowed, so only translation units that compile without them
 * are built here (infra/bvh.cpp, template/camera.h + texture.h + material.h, lib/tiny_obj_loader.h, lib/stb_image.h; see ref_harness.cpp).
 * infra/bvh.cpp's `#include "precomp.h"` resolves to this file because -Iref_build precedes the reference paths.
 */
#pragma once
#include <chrono>
#include <fstream>
#include <vector>
#include <list>
#include <string>
#include <thread>
#include <math.h>
#include <algorithm>
#include <assert.h>
#include <cstring>
#include <immintrin.h>
typedef unsigned char uchar;
typedef unsigned int uint;
typedef unsigned short ushort;
using namespace std;
#define ALIGN( x ) __attribute__( ( aligned( x ) ) )
#define MALLOC64( x ) ( ( x ) == 0 ? 0 : aligned_alloc( 64, ( x ) ) )
#define FREE64( x ) free( x )
#define CHECK_RESULT __attribute__ ((warn_unused_result))
namespace Tmpl8 {}
using namespace Tmpl8;
#include "common.h"      /* reference: template/common.h   */
#include "tmplmath.h"    /* reference: template/tmplmath.h */
#include "ray.h"         /* reference: template/ray.h      */
#include "imgui.h"       /* reference: lib/imgui/imgui.h (helper.h's inline UI helper names ImGui::*) */
#include "helper.h"      /* reference: infra/helper.h (Tri, Vertex) */
