// rtx/SceneManager.h -- SceneManager, source-compatible with the reference's
// src/SceneManager.{h,cpp}: the static create_* factories with their default arguments
// (SceneManager.h:17-25, bodies SceneManager.cpp:137-236), the nine-block upload plumbing
// (SceneManager.cpp:238-276) and the yaw/pitch camera (SceneManager.cpp:43-74).
//
// GLFW keyboard/mouse callbacks (SceneManager.cpp:16-35,76-135) belong to the windowing layer
// that is outside the replaced path; the same camera state is driven programmatically here
// through set_view / press (see the end of the class).
#pragma once

#include <cstring>

#include "GLWrapper.h"
#include "scene.h"

class SceneManager {
public:
    SceneManager(int wind_width, int wind_height, scene_container* scene, GLWrapper* wrapper)
        : scene(scene), wind_width(wind_width), wind_height(wind_height), wrapper(wrapper), position(scene->scene.camera_pos)
    {
    }

    void init() { init_buffers(); }  // SceneManager.cpp:16-35 minus the GLFW callback registration

    void update(float deltaTime)  // SceneManager.cpp:37-41
    {
        update_scene(deltaTime);
        update_buffers();
    }

    static rt_material create_material(glm::vec3 color, int specular, float reflect, float refract = 0.0, glm::vec3 absorb = {},
                                       float diffuse = 0.7, float kd = 0.8, float ks = 0.2)
    {
        rt_material material = {};
        material.color = color;
        material.absorb = absorb;
        material.specular = specular;
        material.reflect = reflect;
        material.refract = refract;
        material.diffuse = diffuse;
        material.kd = kd;
        material.ks = ks;
        return material;
    }
    static rt_sphere create_sphere(glm::vec3 center, float radius, rt_material material, bool hollow = false)
    {
        rt_sphere s = {};
        s.obj = glm::vec4(center, radius);
        s.hollow = hollow;
        s.material = material;
        return s;
    }
    static rt_plane create_plane(glm::vec3 normal, glm::vec3 pos, rt_material material)
    {
        rt_plane p = {};
        p.normal = normal;
        p.pos = pos;
        p.material = material;
        return p;
    }
    static rt_box create_box(glm::vec3 pos, glm::vec3 form, rt_material material)
    {
        rt_box b = {};
        b.form = form;
        b.pos = pos;
        b.mat = material;
        return b;
    }
    static rt_torus create_torus(glm::vec3 pos, glm::vec2 form, rt_material material)
    {
        rt_torus t = {};
        t.form = form;
        t.pos = pos;
        t.mat = material;
        return t;
    }
    static rt_ring create_ring(glm::vec3 pos, float r1, float r2, rt_material material)
    {
        rt_ring r = {};
        r.pos = pos;
        r.mat = material;
        r.r1 = r1 * r1;  // the tracer compares squared radii (rt.frag:382-384)
        r.r2 = r2 * r2;
        return r;
    }
    static rt_light_point create_light_point(glm::vec4 position, glm::vec3 color, float intensity, float linear_k = 0.22f,
                                             float quadratic_k = 0.2f)
    {
        rt_light_point l = {};
        l.intensity = intensity;
        l.pos = position;
        l.color = color;
        l.linear_k = linear_k;
        l.quadratic_k = quadratic_k;
        return l;
    }
    static rt_light_direct create_light_direct(glm::vec3 direction, glm::vec3 color, float intensity)
    {
        rt_light_direct l = {};
        l.intensity = intensity;
        l.direction = direction;
        l.color = color;
        return l;
    }
    static rt_scene create_scene(int width, int height)
    {
        rt_scene s = {};
        s.camera_pos = glm::vec3(0, 0, 0);
        s.canvas_height = height;
        s.canvas_width = width;
        s.bg_color = glm::vec3(0, 0, 0);
        s.reflect_depth = 5;
        return s;
    }

    // ---- programmatic stand-ins for the GLFW input callbacks --------------------------------
    void set_view(float yaw_deg, float pitch_deg)
    {
        yaw = yaw_deg;
        pitch = pitch_deg > 89.0f ? 89.0f : (pitch_deg < -89.0f ? -89.0f : pitch_deg);  // SceneManager.cpp:131-134
    }
    void mouse_moved(double xpos, double ypos)  // SceneManager.cpp:110-135 (glfw_mouse_callback): 0.05 degrees per pixel
    {
        if (firstMouse) {
            lastX = xpos;
            lastY = ypos;
            firstMouse = false;
        }
        float xoffset = static_cast<float>(xpos - lastX);
        float yoffset = static_cast<float>(lastY - ypos);
        lastX = xpos;
        lastY = ypos;
        const float sensitivity = 0.05f;
        xoffset *= sensitivity;
        yoffset *= sensitivity;
        yaw += xoffset;
        pitch += yoffset;
        if (pitch > 89.0f) pitch = 89.0f;
        if (pitch < -89.0f) pitch = -89.0f;
    }
    enum Key { W, A, S, D, SPACE, CTRL, SHIFT, ALT };
    void press(Key k, bool down)
    {
        bool* flags[] = {&w_pressed, &a_pressed, &s_pressed, &d_pressed, &space_pressed, &ctrl_pressed, &shift_pressed, &alt_pressed};
        *flags[k] = down;
    }

private:
    scene_container* scene;
    int wind_width;
    int wind_height;
    GLWrapper* wrapper;

    bool w_pressed = false, a_pressed = false, s_pressed = false, d_pressed = false;
    bool ctrl_pressed = false, shift_pressed = false, space_pressed = false, alt_pressed = false;

    glm::vec3 position;
    glm::vec3 front;
    glm::vec3 right;
    glm::vec3 world_up = glm::vec3(0, 1, 0);
    float yaw = 0;
    float pitch = 0;
    bool firstMouse = true;
    double lastX = 0, lastY = 0;

    GLuint sceneUbo = 0, sphereUbo = 0, planeUbo = 0, surfaceUbo = 0, boxUbo = 0, torusUbo = 0, ringUbo = 0, lightPointUbo = 0,
           lightDirectUbo = 0;

    void update_scene(float deltaTime)  // SceneManager.cpp:43-74
    {
        front.x = glm::sin(glm::radians(yaw)) * glm::cos(glm::radians(pitch));
        front.y = glm::sin(glm::radians(pitch));
        front.z = glm::cos(glm::radians(yaw)) * glm::cos(glm::radians(pitch));
        front = glm::normalize(front);
        right = glm::normalize(glm::cross(-front, world_up));
        scene->scene.quat_camera_rotation = glm::quat(glm::vec3(glm::radians(-pitch), glm::radians(yaw), 0));

        float speed = deltaTime * 3;
        if (shift_pressed) speed *= 3;
        if (alt_pressed) speed /= 6;
        if (w_pressed) position += front * speed;
        if (a_pressed) position -= right * speed;
        if (s_pressed) position -= front * speed;
        if (d_pressed) position += right * speed;
        if (space_pressed) position += world_up * speed;
        if (ctrl_pressed) position -= world_up * speed;
        scene->scene.camera_pos = position;
    }

    template <typename T>
    void init_buffer(GLuint* ubo, const char* name, int bindingPoint, std::vector<T>& v)
    {
        wrapper->init_buffer(ubo, name, bindingPoint, sizeof(T) * v.size(), v.data());
    }
    template <typename T>
    void update_buffer(GLuint ubo, std::vector<T>& v) const
    {
        if (!v.empty()) wrapper->update_buffer(ubo, sizeof(T) * v.size(), v.data());
    }

    void init_buffers()  // SceneManager.cpp:244-255: binding points 0..8
    {
        wrapper->init_buffer(&sceneUbo, "scene_buf", 0, sizeof(rt_scene), nullptr);
        init_buffer(&sphereUbo, "spheres_buf", 1, scene->spheres);
        init_buffer(&planeUbo, "planes_buf", 2, scene->planes);
        init_buffer(&surfaceUbo, "surfaces_buf", 3, scene->surfaces);
        init_buffer(&boxUbo, "boxes_buf", 4, scene->boxes);
        init_buffer(&torusUbo, "toruses_buf", 5, scene->toruses);
        init_buffer(&ringUbo, "rings_buf", 6, scene->rings);
        init_buffer(&lightPointUbo, "lights_point_buf", 7, scene->lights_point);
        init_buffer(&lightDirectUbo, "lights_direct_buf", 8, scene->lights_direct);
    }
    void update_buffers() const  // SceneManager.cpp:266-276: the directional-light block is never refreshed (trap T19)
    {
        wrapper->update_buffer(sceneUbo, sizeof(rt_scene), &scene->scene);
        update_buffer(sphereUbo, scene->spheres);
        update_buffer(planeUbo, scene->planes);
        update_buffer(surfaceUbo, scene->surfaces);
        update_buffer(boxUbo, scene->boxes);
        update_buffer(torusUbo, scene->toruses);
        update_buffer(ringUbo, scene->rings);
        update_buffer(lightPointUbo, scene->lights_point);
    }
};
