// rtx/Surface.h -- SurfaceFactory, source-compatible with the reference's src/Surface.h:7-97.
//
// A quadric is stored as the coefficients of  a x^2 + b y^2 + c z^2 + d z + e y + f = 0  in the
// surface's own frame (rt.frag:67-79). The nine canonical shapes below differ only in which
// coefficients are non-zero; semi-axes enter as 1/axis^2 computed with powf(axis, -2) like the
// reference, so the block bytes match the golden dumps bit for bit.
#pragma once
#include <cmath>
#include <cstring>

#include "scene.h"

class SurfaceFactory {
    // coefficient of a squared term for semi-axis s
    static float inv_sq(float s) { return powf(s, -2); }
    static rt_surface blank(const rt_material& material)
    {
        rt_surface s = {};
        s.mat = material;
        return s;
    }

public:
    // x^2/a^2 + y^2/b^2 + z^2/c^2 = 1
    static rt_surface GetEllipsoid(float a, float b, float c, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.c = inv_sq(c); s.f = -1;
        return s;
    }
    // x^2/a^2 + y^2/b^2 = z
    static rt_surface GetEllipticParaboloid(float a, float b, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.d = -1;
        return s;
    }
    // x^2/a^2 - y^2/b^2 = z
    static rt_surface GetHyperbolicParaboloid(float a, float b, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = -inv_sq(b); s.d = -1;
        return s;
    }
    // x^2/a^2 + y^2/b^2 - z^2/c^2 = 1
    static rt_surface GetEllipticHyperboloidOneSheet(float a, float b, float c, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.c = -inv_sq(c); s.f = -1;
        return s;
    }
    // x^2/a^2 + y^2/b^2 - z^2/c^2 = -1
    static rt_surface GetEllipticHyperboloidTwoSheets(float a, float b, float c, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.c = -inv_sq(c); s.f = 1;
        return s;
    }
    // x^2/a^2 + y^2/b^2 - z^2/c^2 = 0
    static rt_surface GetEllipticCone(float a, float b, float c, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.c = -inv_sq(c);
        return s;
    }
    // x^2/a^2 + y^2/b^2 = 1
    static rt_surface GetEllipticCylinder(float a, float b, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = inv_sq(b); s.f = -1;
        return s;
    }
    // x^2/a^2 - y^2/b^2 = 1
    static rt_surface GetHyperbolicCylinder(float a, float b, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = inv_sq(a); s.b = -inv_sq(b); s.f = -1;
        return s;
    }
    // x^2 + 2 a y = 0
    static rt_surface GetParabolicCylinder(float a, rt_material material)
    {
        rt_surface s = blank(material);
        s.a = 1; s.e = 2 * a;
        return s;
    }
};
