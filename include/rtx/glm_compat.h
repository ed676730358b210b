// rtx/glm_compat.h -- the handful of GLM types/functions the scene-description API needs.
//
// The reference's scene.h / SceneManager / main.cpp describe scenes with glm::vec2/3/4 and
// glm::quat (GLM 0.9.9.7, vendored under external_sources/glm in the reference tree). Only a
// small part of GLM is used on that path; this header re-implements exactly that part so that
// scene-description code written against the reference API compiles unchanged, and produces
// byte-identical uniform-block contents (pinned by tests/golden/*.bin):
//
//   * quat storage order x,y,z,w; constructor order (w,x,y,z)   [glm/detail/type_quat.hpp:45-58,91]
//   * quat(vec3 euler)                                            [glm/detail/type_quat.inl:204-213]
//   * quat *= quat (Hamilton product, this exact term order)      [glm/detail/type_quat.inl:282-292]
//   * quat * vec3, vec3 * quat                                    [glm/detail/type_quat.inl:343-356]
//   * angleAxis                                                   [glm/ext/quaternion_trigonometric.inl:27-33]
//   * radians                                                     [glm/detail/func_trigonometric.inl:9-14]
//
// If the real GLM is wanted instead, define RTX_USE_SYSTEM_GLM before including any rtx header.
#pragma once

#ifdef RTX_USE_SYSTEM_GLM
#include <glm/glm.hpp>
#include <glm/gtc/quaternion.hpp>
#else

#include <cmath>
#include <cstddef>

namespace rtxm {

struct vec2 {
    float x, y;
    vec2() = default;
    template <class A, class B> vec2(A a, B b) : x(static_cast<float>(a)), y(static_cast<float>(b)) {}
    explicit vec2(float s) : x(s), y(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

struct vec3 {
    float x, y, z;
    vec3() = default;
    template <class A, class B, class C>
    vec3(A a, B b, C c) : x(static_cast<float>(a)), y(static_cast<float>(b)), z(static_cast<float>(c)) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3& operator+=(const vec3& o) { x += o.x; y += o.y; z += o.z; return *this; }
    vec3& operator-=(const vec3& o) { x -= o.x; y -= o.y; z -= o.z; return *this; }
    vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
};

struct vec4 {
    float x, y, z, w;
    vec4() = default;
    template <class A, class B, class C, class D>
    vec4(A a, B b, C c, D d)
        : x(static_cast<float>(a)), y(static_cast<float>(b)), z(static_cast<float>(c)), w(static_cast<float>(d)) {}
    template <class D> vec4(const vec3& v, D d) : x(v.x), y(v.y), z(v.z), w(static_cast<float>(d)) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

inline vec3 operator+(const vec3& a, const vec3& b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3& a, const vec3& b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3& a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, const vec3& a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(const vec3& a, const vec3& b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3& a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline bool operator==(const vec3& a, const vec3& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b)
{
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline float length(const vec3& a) { return std::sqrt(dot(a, a)); }
// glm::normalize(vec3) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x)
inline vec3 normalize(const vec3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }

inline float radians(float degrees) { return degrees * 0.01745329251994329576923690768489f; }
inline float sin(float a) { return std::sin(a); }
inline float cos(float a) { return std::cos(a); }

struct quat {
    float x, y, z, w;  // memory order matches the shader's vec4 (x,y,z,w)
    quat() = default;
    // GLM constructor order: (w, x, y, z)
    template <class A, class B, class C, class D>
    quat(A w_, B x_, C y_, D z_)
        : x(static_cast<float>(x_)), y(static_cast<float>(y_)), z(static_cast<float>(z_)), w(static_cast<float>(w_)) {}
    quat(float s, const vec3& v) : x(v.x), y(v.y), z(v.z), w(s) {}
    // Euler angles (pitch, yaw, roll) -> quaternion
    explicit quat(const vec3& eulerAngle)
    {
        const vec3 h = eulerAngle * 0.5f;
        const vec3 c(std::cos(h.x), std::cos(h.y), std::cos(h.z));
        const vec3 s(std::sin(h.x), std::sin(h.y), std::sin(h.z));
        w = c.x * c.y * c.z + s.x * s.y * s.z;
        x = s.x * c.y * c.z - c.x * s.y * s.z;
        y = c.x * s.y * c.z + s.x * c.y * s.z;
        z = c.x * c.y * s.z - s.x * s.y * c.z;
    }
    quat& operator*=(const quat& r)
    {
        const quat p(*this);
        const quat q(r);
        w = p.w * q.w - p.x * q.x - p.y * q.y - p.z * q.z;
        x = p.w * q.x + p.x * q.w + p.y * q.z - p.z * q.y;
        y = p.w * q.y + p.y * q.w + p.z * q.x - p.x * q.z;
        z = p.w * q.z + p.z * q.w + p.x * q.y - p.y * q.x;
        return *this;
    }
};

inline quat operator*(const quat& a, const quat& b) { return quat(a) *= b; }
inline float dot(const quat& a, const quat& b)
{
    // glm compute_dot<qua> (detail/type_quat.inl:16-22): (w*w + x*x) + (y*y + z*z)
    return (a.w * b.w + a.x * b.x) + (a.y * b.y + a.z * b.z);
}
inline quat conjugate(const quat& q) { return quat(q.w, -q.x, -q.y, -q.z); }
inline quat inverse(const quat& q)
{
    const quat c = conjugate(q);
    const float d = dot(q, q);
    return quat(c.w / d, c.x / d, c.y / d, c.z / d);
}
inline vec3 operator*(const quat& q, const vec3& v)
{
    const vec3 QuatVector(q.x, q.y, q.z);
    const vec3 uv(cross(QuatVector, v));
    const vec3 uuv(cross(QuatVector, uv));
    return v + ((uv * q.w) + uuv) * 2.0f;
}
inline vec3 operator*(const vec3& v, const quat& q) { return inverse(q) * v; }

inline quat angleAxis(float angle, const vec3& v)
{
    const float a = angle;
    const float s = std::sin(a * 0.5f);
    return quat(std::cos(a * 0.5f), v * s);
}

}  // namespace rtxm

namespace glm = rtxm;

#endif  // RTX_USE_SYSTEM_GLM
